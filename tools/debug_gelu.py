import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests"), os.path.join(ROOT, "tests/golden")]
import torch
import ofq_oracle as O
from ofq_amd import ops
from detgen import det_normalish, det_uniform
T = torch.from_numpy
B, N, C, bits = 4, 198, 256, 2
lo, hi = 0, 3
x = T(det_normalish((B, N, C), 11, 4.0))
b4 = torch.zeros(C); baft = torch.zeros(C)
s = O.lsq_token_init(torch.nn.functional.gelu(x), bits, True).contiguous()
xr = x.clone().requires_grad_(True); sr = s.clone().requires_grad_(True); b4r = b4.clone().requires_grad_(True); bar = baft.clone().requires_grad_(True)
y = O.lsq_token(torch.nn.functional.gelu(xr) + b4r, sr, bits, True) + bar
gy = T(det_uniform((B, N, C), 15, -1.0, 1.0))
(y * gy).sum().backward()
geom = ops.LsqGeom(B, N, C, C, 0, lo, hi, B * C, prologue=1)
yg, _ = ops.lsq_fwd(x.cuda(), s.cuda(), b4.cuda(), baft.cuda(), geom)
dx, ds, db4, dbaft = ops.lsq_bwd(gy.cuda(), x.cuda(), s.cuda(), b4.cuda(), geom)
yg = yg.cpu().reshape(B, N, C); dx = dx.cpu().reshape(B, N, C)
print("y mismatches", (yg != y.detach()).sum().item())
d = (dx - xr.grad).abs()
print("dx max abs err", d.max().item(), "ref max", xr.grad.abs().max().item())
idx = (d > 1e-4).nonzero()
print("n bad", len(idx))
for i in idx[:10]:
    i = tuple(i.tolist())
    print("h=%.6f gelu=%.6e y_ref=%.4f y_gpu=%.4f g=%.4f dx_ref=%.6f dx_gpu=%.6f s=%.4f" % (x[i], torch.nn.functional.gelu(x[i]), y[i], yg[i], gy[i], xr.grad[i], dx[i], s[i[1]]))
print("db4 err", (db4.cpu() - b4r.grad).abs().max().item(), b4r.grad.abs().max().item())
print("ds err", (ds.cpu() - sr.grad).abs().max().item(), sr.grad.abs().max().item())
