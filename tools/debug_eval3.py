#!/usr/bin/env python3
"""Where inside block 0 does the product leave the oracle on a fresh full-width DeiT-T?"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import torch
import ofq_oracle as O
from ofq_amd import engine, functional as F_ofq

def rel(a, b):
    a, b = a.detach().cpu().double(), b.detach().double()
    return float((a - b).norm() / b.norm()), float((a - b).abs().max() / b.abs().max())

torch.manual_seed(0)
depth, bits = 12, 3
model = engine.build_student("deit_tiny_distilled_patch16_224", bits, bits, qk_reparam=True, depth=depth).cuda()
x = torch.randn(4, 3, 224, 224, device="cuda")
engine.setup_alpha(model, x)
model.eval()
sd = {k: v.detach().cpu() for k, v in model.state_dict().items()}
H, C = 3, 192
with torch.no_grad():
    t = O.qconv_patch_embed(x.cpu(), O._sub(sd, "patch_embed.proj."), 16)
    B = t.shape[0]
    t0 = torch.cat((sd["cls_token"].expand(B, -1, -1), sd["dist_token"].expand(B, -1, -1), t), dim=1) + sd["pos_embed"]
    p = O._sub(sd, "blocks.0.")
    a = torch.nn.functional.layer_norm(t0, (C,), p["norm1.weight"], p["norm1.bias"], 1e-6)
    ya = O.qattention_qkr(a, O._sub(p, "attn."), H, bits, bits)
    t1 = t0 + ya
    m = torch.nn.functional.layer_norm(t1, (C,), p["norm2.weight"], p["norm2.bias"], 1e-6)
    ym = O.qmlp(m, O._sub(p, "mlp."), bits, bits)
    # product, op by op on the ORACLE's inputs (so errors do not compound)
    g0 = model._tokens(x)
    print("tokens (stem + pos)         ", rel(g0, t0))
    blk = model.blocks[0]
    F_ofq.BULK_WQK = False
    ga, _ = blk.attn(a.cuda())
    print("attention on oracle's input ", rel(ga, ya))
    gm = blk.mlp(m.cuda())
    print("mlp on oracle's input       ", rel(gm, ym))
    # inside the attention: v branch, qkx branch
    pa = O._sub(p, "attn.")
    from ofq_amd.quantization.modules import qlinear as ql
    for flag in (True, False):
        ql.USE_CODE_GEMM = flag
        ga2, _ = blk.attn(a.cuda())
        print("   attention, code GEMMs=%s  " % flag, rel(ga2, ya))
    ql.USE_CODE_GEMM = True
    # the oracle's own pieces
    xq = O.lsq_token(a + pa["quant_x_4_qkv.move_b4.bias"], pa["quant_x_4_qkv.input_quant_fn.s"], bits, False) + pa["quant_x_4_qkv.move_aft.bias"]
    xin = blk.attn.quant_x_4_qkv
    gxq = xin(a.cuda())
    print("   x_hat                     ", rel(gxq, xq))
    Wv = O.statsq(pa["v.weight"], bits)[0]
    gWv = blk.attn.v_quant(blk.attn.v.weight)
    print("   StatsQ(v.weight)          ", rel(gWv, Wv))
    Wq, Wk = pa["q.weight"], pa["k.weight"]
    d = C // H
    Wqk = torch.cat([Wq[h * d:(h + 1) * d].t() @ Wk[h * d:(h + 1) * d] for h in range(H)], 0)
    gWqk = F_ofq.WqkFn.apply(blk.attn.q.weight, blk.attn.k.weight, H)
    print("   W_qk = Wq^T Wk            ", rel(gWqk, Wqk), " |W_qk| max %.3e" % float(Wqk.abs().max()))
    sW = O.statsq(Wqk, bits)
    gsW = blk.attn.qk_quant(gWqk)
    print("   StatsQ(W_qk)              ", rel(gsW, sW[0]))
    print("   StatsQ(oracle W_qk) on GPU", rel(blk.attn.qk_quant(Wqk.cuda()), sW[0]))
    # ---- inside the MLP
    pm = O._sub(p, "mlp.")
    h_ref = O.qlinear(m, O._sub(pm, "fc1."), bits, bits, unsigned=False)
    mlp = blk.mlp
    ql.FUSE_NEXT_CODES = False
    h_got = mlp.fc1(m.cuda())
    print("   fc1 output                ", rel(h_got, h_ref), " |h| max %.3f" % float(h_ref.abs().max()))
    g_ref = torch.nn.functional.gelu(h_ref)
    p2 = O._sub(pm, "fc2.")
    s2 = p2["input_quant_fn.s"]
    lo, hi = O.lsq_bounds(bits, True)
    gs = 1.0 / ((hi * (m.shape[0] * 768)) ** 0.5)
    lv_ref = O.lsq_levels(g_ref + p2["move_b4.bias"], s2.view(1, -1, 1), lo, hi, gs)
    xq, codes, geom = mlp.fc2.input_quant_fn.quant(h_ref.cuda(), mlp.fc2.move_b4.bias, mlp.fc2.move_aft.bias, prologue=1,
                                                   want_codes=True, need_values=True)
    lv_got = codes.view(lv_ref.shape).cpu().float()
    print("   fc2 input levels on oracle's h: mismatching share %.3e   (s: min %.3e max %.3e, levels used up to %d)"
          % (float((lv_got != lv_ref).float().mean()), float(s2.min()), float(s2.max()), int(lv_ref.max())))
    bad = (lv_got != lv_ref).nonzero()
    if len(bad):
        i = tuple(bad[0].tolist())
        print("      first mismatch at", i, "h", float(h_ref[i]), "gelu", float(g_ref[i]), "s", float(s2[i[1]]), "v", float(g_ref[i] / s2[i[1]]),
              "got", float(lv_got[i]), "want", float(lv_ref[i]))
    y2_ref = O.qlinear(g_ref, p2, bits, bits, unsigned=True)
    y2_got = mlp.fc2(g_ref.cuda())
    print("   fc2 on oracle's gelu(h)   ", rel(y2_got, y2_ref))
    # ---- inside fc1
    p1 = O._sub(pm, "fc1.")
    W_ref, s_w, _ = O.statsq(p1["weight"], bits)
    W_got = mlp.fc1.statsq_fn(mlp.fc1.weight)
    print("   StatsQ(fc1.weight)        ", rel(W_got, W_ref), " differing elements: %d" % int((W_got.cpu() != W_ref).sum()))
    x_ref = O.lsq_token(m + p1["move_b4.bias"], p1["input_quant_fn.s"], bits, False) + p1["move_aft.bias"]
    x_got = mlp.fc1.input_quant_fn.quant(m.cuda(), mlp.fc1.move_b4.bias, mlp.fc1.move_aft.bias)
    print("   fc1 x_hat                 ", rel(x_got, x_ref), " differing elements: %d of %d" % (int((x_got.cpu() != x_ref).sum()), x_ref.numel()))
    y_ref = torch.nn.functional.linear(x_ref, W_ref) + p1["bias"]
    print("   oracle qlinear == F.linear(x_hat, W_hat)?", rel(h_ref, y_ref))
    y_got64 = (x_got.double() @ W_got.double().t() + mlp.fc1.bias.double()).cpu()
    print("   product fc1 vs fp64 product of ITS OWN x_hat, W_hat", rel(h_got, y_got64))
    print("   oracle  fc1 vs fp64 product of ITS OWN x_hat, W_hat", rel(h_ref, (x_ref.double() @ W_ref.double().t() + p1["bias"].double())))
    ql.USE_CODE_GEMM = False
    h_fp = mlp.fc1(m.cuda())
    ql.USE_CODE_GEMM = True
    print("   product fc1 (fp32 GEMM path) vs oracle", rel(h_fp, h_ref))
    Wd = (W_got.cpu() - W_ref).abs()
    big = (Wd > 1e-4).nonzero()
    print("   StatsQ level mismatches: %d" % len(big))
    Wt = p1["weight"]
    s_cpu = 2 * Wt.abs().mean(1)
    s_gpu = mlp.fc1.statsq_fn.s
    s64 = (2 * Wt.double().abs().mean(1))
    print("   scale: GPU vs CPU-fp32 differing rows %d / %d; GPU == round(fp64) rows %d; CPU == round(fp64) rows %d"
          % (int((s_gpu != s_cpu).sum()), len(s_cpu), int((s_gpu == s64.float()).sum()), int((s_cpu == s64.float()).sum())))
    for (r, c) in big[:5].tolist():
        w = float(Wt[r, c])
        for nm, sv in (("cpu", float(s_cpu[r])), ("gpu", float(s_gpu[r]))):
            v = torch.tensor(w) / torch.tensor(sv)
            cc = torch.clamp(v, -1.0, 1.0 - 1e-6)
            print("      [%d,%d] W=%.9g s_%s=%.9g  v=%.9g  c*n-0.5=%.9g  L=%g" % (r, c, w, nm, sv, float(v), float(cc * 4 - 0.5), float(torch.round(cc * 4 - 0.5))))
        print("      got %.9g want %.9g" % (float(W_got[r, c]), float(W_ref[r, c])))
    # all 48 StatsQ weight matrices of the model: how many level flips in total?
    tot = 0
    for n_, mod in model.named_modules():
        if hasattr(mod, "statsq_fn") and hasattr(mod, "weight"):
            a_ = mod.statsq_fn(mod.weight).cpu()
            b_ = O.statsq(mod.weight.detach().cpu(), bits)[0]
            nf = int(((a_ - b_).abs() > 1e-4).sum())
            tot += nf
    print("   level flips over all QLinear weights of the model: %d" % tot)
