import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "oracle")]
import torch, torch.nn as nn
from functools import partial
from types import SimpleNamespace
import ofq_oracle as O
from ofq_amd import engine
from ofq_amd.deit import DistilledVisionTransformer
from ofq_amd.quantization.utils import KDLossSoftandHard
torch.manual_seed(0)
depth, dim, heads, ncls, B = int(os.environ.get("DEPTH", 2)), int(os.environ.get("DIM", 64)), 2, 10, int(os.environ.get("B", 4))
mlp_ratio = int(os.environ.get("MLP", 4))
model = DistilledVisionTransformer(img_size=224, patch_size=16, embed_dim=dim, depth=depth, num_heads=heads,
                                   mlp_ratio=mlp_ratio, qkv_bias=True, num_classes=ncls,
                                   norm_layer=partial(nn.LayerNorm, eps=1e-6), act_layer=nn.GELU)
with torch.no_grad():
    for p in model.parameters():
        if p.dim() >= 2:
            p.mul_(4.0)
args = SimpleNamespace(qmodules=engine.default_qmodules(depth), wq_mode="statsq", wq_enable=True, wq_bitw=2,
                       aq_enable=True, aq_mode="lsq", aq_bitw=2, wq_per_channel=True, aq_per_channel=True,
                       model_type="deit", pretrained_initialized=True, qk_reparam=True, qk_reparam_type=0)
model = engine.get_qat_model(model, args).cuda()
img = torch.randn(B, 3, 224, 224, device="cuda")
tgt = torch.randint(0, ncls, (B,), device="cuda")
soft = torch.randn(B, ncls, device="cuda")
engine.setup_alpha(model, img)
model.train()
(c, d), _ = model(img)
loss = KDLossSoftandHard()((c, d), tgt, soft)
loss.backward()
sd = {k: v.detach().cpu().clone() for k, v in model.state_dict().items()}
leaves = {k: (v.requires_grad_(True) if v.dtype.is_floating_point and "clip_val" not in k and "signed" not in k else v) for k, v in sd.items()}
cfg = dict(depth=depth, num_heads=heads, patch=16, wbits=2, abits=2, qkr=True)
co, do = O.deit_forward(img.cpu(), leaves, cfg, training=True)
lo = O.kd_loss_soft_and_hard(co, do, tgt.cpu(), soft.cpu())
lo.backward()
def rel(a, b):
    return float((a.detach().cpu().double() - b.detach().double()).abs().max() / (b.detach().double().abs().max() + 1e-30))
for n, p in reversed(list(model.named_parameters())):
    if p.grad is None or leaves[n].grad is None:
        print("%-50s NONE" % n); continue
    print("%-50s %.2e   |ref|max %.2e" % (n, rel(p.grad, leaves[n].grad), float(leaves[n].grad.abs().max())))

# ---- capture h = fc1 output and dh for the last block in both implementations
print("---- fc1 output / grad comparison (last block)")
cap = {}
blk = model.blocks[depth - 1].mlp
def fh(mod, inp, out):
    cap["h_gpu"] = out
    out.register_hook(lambda g: cap.__setitem__("dh_gpu", g))
hd = blk.fc1.register_forward_hook(fh)
model.zero_grad()
(c, d), _ = model(img)
KDLossSoftandHard()((c, d), tgt, soft).backward()
hd.remove()
orig_qmlp = O.qmlp
calls = []
def qmlp_cap(x, p, wb, ab):
    h = O.qlinear(x, O._sub(p, "fc1."), wb, ab, unsigned=False)
    h.retain_grad()
    calls.append(h)
    h2 = torch.nn.functional.gelu(h)
    return O.qlinear(h2, O._sub(p, "fc2."), wb, ab, unsigned=True)
O.qmlp = qmlp_cap
for v in leaves.values():
    if v.grad is not None: v.grad = None
co, do = O.deit_forward(img.cpu(), leaves, cfg, training=True)
O.kd_loss_soft_and_hard(co, do, tgt.cpu(), soft.cpu()).backward()
h_ref = calls[-1]
hg = cap["h_gpu"].detach().cpu(); dhg = cap["dh_gpu"].detach().cpu()
print("h err", rel(hg, h_ref), " dh err", rel(dhg, h_ref.grad))
dd = (dhg - h_ref.grad).abs()
idx = (dd > 1e-3 * h_ref.grad.abs().max()).nonzero()
print("n bad", len(idx), "of", dd.numel(), " h std", float(h_ref.std()))
s2 = leaves["blocks.%d.mlp.fc2.input_quant_fn.s" % (depth - 1)].detach()
b42 = leaves["blocks.%d.mlp.fc2.move_b4.bias" % (depth - 1)].detach()
for i in idx[:12]:
    i = tuple(i.tolist())
    hv = h_ref[i].item()
    print("h=%.7f (gpu %.7f) gelu=%.6e s=%.5f v=%.6f dh_ref=%.6e dh_gpu=%.6e" % (hv, hg[i].item(), torch.nn.functional.gelu(h_ref[i]).item(), s2[i[1]].item(), (torch.nn.functional.gelu(h_ref[i]).item() + b42[i[2]].item()) / s2[i[1]].item(), h_ref.grad[i].item(), dhg[i].item()))
