#!/usr/bin/env python3
"""Is the depth-12 deviation from the oracle chaos (rounding ties amplified by 12 quantised blocks) or a defect?
(1) product vs product under a 1e-7 relative input perturbation; (2) oracle vs oracle under the same perturbation;
(3) product with per-block W_qk instead of the batched one; (4) per-block divergence from the oracle."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import torch
import ofq_oracle as O
from ofq_amd import engine, functional as F_ofq

def rel(a, b):
    return float((a.double() - b.double()).norm() / b.double().norm())

torch.manual_seed(0)
depth, bits = 12, 3
model = engine.build_student("deit_tiny_distilled_patch16_224", bits, bits, qk_reparam=True, depth=depth).cuda()
x = torch.randn(4, 3, 224, 224, device="cuda")
engine.setup_alpha(model, x)
model.eval()
sd = {k: v.detach().cpu() for k, v in model.state_dict().items()}
cfg = dict(depth=depth, num_heads=3, patch=16, wbits=bits, abits=bits, qkr=True)
with torch.no_grad():
    y0, _ = model(x)
    y1, _ = model(x * (1 + 1e-7))
    r0 = O.deit_forward(x.cpu(), sd, cfg, training=False)
    r1 = O.deit_forward((x * (1 + 1e-7)).cpu(), sd, cfg, training=False)
    F_ofq.BULK_WQK = False
    y2, _ = model(x)
    F_ofq.BULK_WQK = True
print("product vs oracle            %.3e" % rel(y0.cpu(), r0))
print("product vs product(x*(1+1e-7)) %.3e" % rel(y1, y0))
print("oracle  vs oracle (x*(1+1e-7)) %.3e" % rel(r1, r0))
print("product (per-block W_qk) vs product %.3e" % rel(y2, y0))
# per-block activations: forward_features returns the block outputs; the oracle trunk is re-run to each depth
def oracle_trunk(d):
    with torch.no_grad():
        H = cfg["num_heads"]
        t = O.qconv_patch_embed(x.cpu(), O._sub(sd, "patch_embed.proj."), 16)
        B = t.shape[0]
        t = torch.cat((sd["cls_token"].expand(B, -1, -1), sd["dist_token"].expand(B, -1, -1), t), dim=1) + sd["pos_embed"]
        C = t.shape[-1]
        outs = []
        for i in range(d):
            p = O._sub(sd, "blocks.%d." % i)
            a = torch.nn.functional.layer_norm(t, (C,), p["norm1.weight"], p["norm1.bias"], 1e-6)
            t = t + O.qattention_qkr(a, O._sub(p, "attn."), H, bits, bits)
            m = torch.nn.functional.layer_norm(t, (C,), p["norm2.weight"], p["norm2.bias"], 1e-6)
            t = t + O.qmlp(m, O._sub(p, "mlp."), bits, bits)
            outs.append(t)
    return outs

ref = oracle_trunk(depth)
from ofq_amd.quantization.modules import qlinear as ql
def show(tag):
    with torch.no_grad():
        feats = model.forward_features(x)[3]
    print(tag, " ".join("%.1e" % rel(f.cpu(), r) for f, r in zip(feats, ref)))
show("default              ")
ql.FUSE_NORM_QUANT = False; show("no norm+quant fusion "); ql.FUSE_NORM_QUANT = True
ql.FUSE_NEXT_CODES = False; show("no epilogue codes    "); ql.FUSE_NEXT_CODES = True
sites = ql.RECOMPUTE_SITES; ql.RECOMPUTE_SITES = frozenset(); show("no recompute         "); ql.RECOMPUTE_SITES = sites
ql.USE_CODE_GEMM = False; show("fp32 GEMMs           "); ql.USE_CODE_GEMM = True
