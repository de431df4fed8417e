"""Full-depth parity: the 12-block models of BASELINE.json's configurations (DeiT-T W4A4 plain attention, DeiT-S W2A2 QKR)
on the HIP path against the CPU oracle, at the real dimensions, two images.

What can and cannot be asked of a 12-block 2/4-bit network.  Every block re-rounds its activations, so the network is a
discontinuous function of its inputs: a value within fp32 rounding noise of a rounding tie takes one level in one
implementation and the neighbouring level in another (different summation order), that moves one token's features by about
a level step, the next block's quantisers turn the disturbance into further flips, and after a few blocks the two runs have
decorrelated by percents.  This is a property of the function, not of an implementation: the reference's own arithmetic
run in fp64 instead of fp32 diverges from its fp32 self in exactly the same way (measured below, `self_div`).  With ~25 M
quantised values per forward and a flip rate of ~4e-7 per value (tests/golden/prodcases.py: one production-size module in
three has one), a flip-free depth-12 forward does not exist.  So:

  * per block, TEACHER-FORCED: each HIP block gets the oracle's input of that block; output, input gradient and every
    parameter gradient must match the oracle's block at 1e-3.  Tokens touched by a flipped level are counted and bounded,
    not hidden: the other tokens must agree to 1e-3 (measured: 7e-8) and the block output overall to 2e-2; at least
    six of the twelve blocks must be flip-free (measured: 9 and 10), i.e. agree in the output AND every parameter gradient at 1e-3.  This pins
    every block at real dimensions and depth-specific parameters;
  * end to end, with the oracle's StatsQ scales injected into the HIP path (StatsQuantizer._scale_override, so that the
    weight levels are the oracle's): the HIP logits must be as close to the oracle's as the oracle's fp64 run is
    (`e2e <= 3 * self_div + 1e-3`), and the first blocks -- before any flip has been amplified -- must agree to 1e-3."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

import ofq_oracle as O
from util import rel_err

pytestmark = pytest.mark.gpu
TOL = 1e-3


def _build(model_name, wb, ab, qkr, seed=0):
    from ofq_amd import engine
    model = engine.build_student(model_name, wb, ab, qk_reparam=qkr, seed=42).cuda()
    g = torch.Generator(device="cpu").manual_seed(seed)
    with torch.no_grad():                      # biases / offsets off zero (zero-initialised biases put pre-activations ON
        for n, p in model.named_parameters():  # the unsigned quantisers' clamp edge, where fp32 noise decides the mask)
            if p.dim() == 1 and "norm" not in n:
                p.add_(0.05 * torch.randn(p.shape, generator=g).cuda())
    img = torch.randn(2, 3, 224, 224, generator=g).cuda()
    engine.setup_alpha(model, img)
    return model, img


def _oracle_chain(img, sd, cfg, dtype):
    """O.deit_forward (deit.py:32-67) unrolled so that every block's input and output are kept."""
    sd = {k: (v.to(dtype) if v.dtype.is_floating_point else v) for k, v in sd.items()}
    H, wb, ab = cfg["num_heads"], cfg["wbits"], cfg["abits"]
    x = O.qconv_patch_embed(img.to(dtype), O._sub(sd, "patch_embed.proj."), cfg["patch"])
    B = x.shape[0]
    x = torch.cat((sd["cls_token"].expand(B, -1, -1), sd["dist_token"].expand(B, -1, -1), x), dim=1) + sd["pos_embed"]
    C = x.shape[-1]
    attn_fn = O.qattention_qkr if cfg["qkr"] else O.qattention
    xs = [x]
    for i in range(cfg["depth"]):
        x = _oracle_block(x, O._sub(sd, "blocks.%d." % i), cfg, attn_fn)
        xs.append(x)
    h = F.layer_norm(x, (C,), sd["norm.weight"], sd["norm.bias"], 1e-6)
    logits = (O.qhead(h[:, 0], O._sub(sd, "head.")) + O.qhead(h[:, 1], O._sub(sd, "head_dist."))) / 2
    return xs, logits


def _oracle_block(x, p, cfg, attn_fn):
    C = x.shape[-1]
    h = F.layer_norm(x, (C,), p["norm1.weight"], p["norm1.bias"], 1e-6)
    x = x + attn_fn(h, O._sub(p, "attn."), cfg["num_heads"], cfg["wbits"], cfg["abits"])
    h = F.layer_norm(x, (C,), p["norm2.weight"], p["norm2.bias"], 1e-6)
    return x + O.qmlp(h, O._sub(p, "mlp."), cfg["wbits"], cfg["abits"])


def _l2(a, b):
    a, b = a.detach().double().cpu().reshape(-1), b.detach().double().cpu().reshape(-1)
    return float((a - b).norm() / (b.norm() + 1e-30))


def _token_flips(y, yo):
    """(number of tokens with an element off by more than 1e-3 of the tensor's max, l2 over the other tokens)."""
    y, yo = y.detach().double().cpu(), yo.detach().double().cpu()
    e = (y - yo).abs().reshape(-1, y.shape[-1]).max(1).values / (yo.abs().max() + 1e-30)
    bad = e > TOL
    keep = ~bad
    yk, yok = y.reshape(-1, y.shape[-1])[keep], yo.reshape(-1, y.shape[-1])[keep]
    return int(bad.sum()), float((yk - yok).norm() / (yok.norm() + 1e-30))


CONFIGS = [("deit_tiny_distilled_patch16_224", 4, 4, False, 3),        # BASELINE configs[0/1]: DeiT-T W4A4, plain attention
           ("deit_small_distilled_patch16_224", 2, 2, True, 6)]        # configs[2]: DeiT-S W2A2, QKR


@pytest.mark.parametrize("model_name,wb,ab,qkr,H", CONFIGS)
def test_every_block_of_the_full_depth_model_teacher_forced(model_name, wb, ab, qkr, H):
    model, img = _build(model_name, wb, ab, qkr)
    model.train()
    sd = {k: v.detach().cpu() for k, v in model.state_dict().items()}
    cfg = dict(depth=12, num_heads=H, patch=16, wbits=wb, abits=ab, qkr=qkr)
    with torch.no_grad():
        xs, _ = _oracle_chain(img.cpu(), sd, cfg, torch.float32)
    attn_fn = O.qattention_qkr if qkr else O.qattention
    gen = torch.Generator().manual_seed(7)
    clean, rows = 0, []
    for i, blk in enumerate(model.blocks):
        up = torch.randn(xs[i].shape, generator=gen)
        # oracle block with autograd
        p = {k: v.clone().requires_grad_(v.dtype.is_floating_point and "clip_val" not in k)
             for k, v in O._sub(sd, "blocks.%d." % i).items()}
        xo = xs[i].clone().requires_grad_(True)
        yo = _oracle_block(xo, p, cfg, attn_fn)
        (yo * up).sum().backward()
        # HIP block on the same input
        blk.zero_grad(set_to_none=True)
        xh = xs[i].cuda().requires_grad_(True)
        yh, _ = blk(xh)
        (yh * up.cuda()).sum().backward()
        flips, l2_rest = _token_flips(yh, yo)
        ntok = yo.shape[0] * yo.shape[1]
        # one flipped level of v or qkx reaches every token that attends to it: a single flip can touch a few percent of
        # the tokens (by a few 1e-3 of the output range each)
        assert flips <= ntok // 10 and l2_rest < TOL and _l2(yh, yo) < 2e-2, (i, flips, l2_rest, _l2(yh, yo))
        # gradients: a flipped level changes clip masks and rounding residues of the tokens it touches -- a per-token step
        # gradient ds[n] of a touched token moves by O(1), i.e. by sqrt(touched / 198) of that vector's norm -- and a flip
        # too small to show in the output (< 1e-3 of its range) still does that.  So a block counts as CLEAN when its output
        # and every gradient agree to 1e-3; the others are reported, and bounded on the output only.
        errs = {"dx": _l2(xh.grad, xo.grad)}
        for n, q in blk.named_parameters():
            if q.grad is None:
                continue
            ref = p[n].grad
            assert ref is not None, n
            if "move_" in n and float(ref.abs().max()) < 1e-4 * max(float(v.grad.abs().max()) for k, v in p.items()
                                                                     if "move_" in k and v.grad is not None):
                continue                      # offsets whose gradient is identically zero in exact arithmetic: noise
            errs[n] = _l2(q.grad, ref)
        clean += int(flips == 0 and max(errs.values()) < TOL)
        rows.append((i, flips, l2_rest, _l2(yh, yo), max(errs.values())))
    print("\n%s W%dA%d: block, tokens touched by a flip, l2(y) on the rest, l2(y) overall, worst gradient l2" % (model_name, wb, ab))
    for r in rows:
        print("   %2d  %3d  %.2e  %.2e  %.2e" % r)
    assert clean >= 6, rows           # blocks without any flip: output and every gradient at 1e-3 (measured: 9 and 10 of 12)


@pytest.mark.parametrize("model_name,wb,ab,qkr,H", CONFIGS)
def test_full_depth_end_to_end_with_the_oracles_statsq_scales(model_name, wb, ab, qkr, H):
    from ofq_amd.quantization.quantizer.statsq import StatsQuantizer
    model, img = _build(model_name, wb, ab, qkr, seed=1)
    model.eval()
    sd = {k: v.detach().cpu() for k, v in model.state_dict().items()}
    cfg = dict(depth=12, num_heads=H, patch=16, wbits=wb, abits=ab, qkr=qkr)
    with torch.no_grad():
        xs32, lo32 = _oracle_chain(img.cpu(), sd, cfg, torch.float32)
        xs64, lo64 = _oracle_chain(img.cpu(), sd, cfg, torch.float64)
    # the oracle's StatsQ scales (torch-CPU fp32 row means) into the HIP path
    n_inj = 0
    for name, mod in model.named_modules():
        for qn, wn in (("statsq_fn", "weight"), ("v_quant", "v.weight")):
            qz = getattr(mod, qn, None)
            if isinstance(qz, StatsQuantizer) and (name + "." + wn) in sd:
                W = sd[name + "." + wn]
                qz._scale_override = (2 * W.abs().mean(dim=1)).contiguous()
                n_inj += 1
        qz = getattr(mod, "qk_quant", None)
        if isinstance(qz, StatsQuantizer):
            C = sd[name + ".q.weight"].shape[1]
            Wq, Wk = sd[name + ".q.weight"].reshape(H, -1, C), sd[name + ".k.weight"].reshape(H, -1, C)
            Wqk = (Wq.transpose(-2, -1).contiguous() @ Wk).reshape(H * C, C)
            qz._scale_override = (2 * Wqk.abs().mean(dim=1)).contiguous()
            n_inj += 1
    assert n_inj >= 48
    try:
        with torch.no_grad():
            out, _ = model(img)
            feats = [f.detach().cpu() for f in model.forward_features(img)[3]]       # the outputs of blocks 0 .. 11
    finally:
        for mod in model.modules():
            if isinstance(mod, StatsQuantizer):
                mod._scale_override = None
    e2e = _l2(out, lo32)
    self_div = _l2(lo64, lo32)
    per_block = [_l2(f, x) for f, x in zip(feats, xs32[1:])] if len(feats) == 12 else []
    per_block_self = [_l2(a, b) for a, b in zip(xs64[1:], xs32[1:])]
    print("\n%s W%dA%d end to end: HIP vs oracle %.2e, oracle fp64 vs fp32 %.2e" % (model_name, wb, ab, e2e, self_div))
    print("   per block HIP   :", " ".join("%.1e" % v for v in per_block))
    print("   per block oracle:", " ".join("%.1e" % v for v in per_block_self))
    assert e2e <= 3 * self_div + TOL, (e2e, self_div)
    if per_block:
        assert per_block[0] < max(TOL, 3 * per_block_self[0]), per_block[:3]
