"""Full-depth parity of BASELINE.json's config 4 (Swin-T W3A3, QKR window attention) and of the CGA fine-tune (config 5) at
real dimensions, on the HIP path against the CPU oracle.

Swin-T: all 12 blocks (stages of 2 / 2 / 6 / 2 at dims 96 / 192 / 384 / 768, heads 3 / 6 / 12 / 24, windows of 7 x 7 tokens,
every other block shifted) and the three PatchMerging layers with their quantised `reduction` (swin.py:40-60, :441-470;
swin_attention_and_mlp.py:253-461), two images, TEACHER-FORCED like tests/test_depth12_gpu.py: each HIP block / merging layer
gets the oracle's input of that layer; output, input gradient and every parameter gradient against the oracle's.  The same
caveat holds (a 3-bit 12-block network is discontinuous: a value within fp32 noise of a rounding tie takes the neighbouring
level under another summation order and touches the tokens of its window): tokens touched by a flip are counted and bounded,
the others must agree to 1e-3, and at least half of the blocks must be flip-free in the output and every gradient.

CGA (cga.py:450-469, :953-1013) at DeiT-S size: the freeze masks of all 48 quantised weight matrices against the oracle's
freeze_outside_boundary_weight_idx, bit for bit, and the frozen weights across a real step; the Swin branch (cga.py:957-964,
`reduction` included) on the full Swin-T."""
import pytest
import torch
import torch.nn.functional as F

import ofq_oracle as O
from test_depth12_gpu import _l2, _token_flips, TOL

pytestmark = pytest.mark.gpu

SWIN_T = dict(depths=[2, 2, 6, 2], num_heads=[3, 6, 12, 24], window=[7, 7], patch=4, wbits=3, abits=3, qkr=True)


def _build_swin(seed=0):
    from ofq_amd import engine
    model = engine.build_student("swin_t", 3, 3, qk_reparam=True, seed=42).cuda()
    g = torch.Generator(device="cpu").manual_seed(seed)
    with torch.no_grad():                      # biases / offsets off zero (see tests/test_depth12_gpu.py)
        for n, p in model.named_parameters():
            if p.dim() == 1 and "norm" not in n:
                p.add_(0.05 * torch.randn(p.shape, generator=g).cuda())
    img = torch.randn(2, 3, 224, 224, generator=g).cuda()
    engine.setup_alpha(model, img)
    return model, img


def _oracle_block(x, p, cfg, si, li):
    """one SwinTransformerBlock of O.swin_forward (swin.py:206-230 with the Q-modules)"""
    C = x.shape[-1]
    shift = [0 if li % 2 == 0 else w // 2 for w in cfg["window"]]
    h = F.layer_norm(x, (C,), p["norm1.weight"], p["norm1.bias"], 1e-5)
    x = x + O.swin_window_attention(h, O._sub(p, "attn."), cfg["num_heads"][si], cfg["window"], shift, cfg["wbits"], cfg["abits"],
                                    cfg["qkr"])
    h = F.layer_norm(x, (C,), p["norm2.weight"], p["norm2.bias"], 1e-5)
    return x + O.qmlp(h, O._sub(p, "mlp."), cfg["wbits"], cfg["abits"])


def _oracle_layers(img, sd, cfg):
    """O.swin_forward unrolled: [(kind, prefix, stage, layer, input of that layer)] for every block and merging layer."""
    x = O.qconv_patch_embed_nhwc(img, O._sub(sd, "features.0.0."), cfg["patch"])
    x = F.layer_norm(x, (x.shape[-1],), sd["features.0.2.weight"], sd["features.0.2.bias"], 1e-5)
    out, fi = [], 1
    for si, depth in enumerate(cfg["depths"]):
        for li in range(depth):
            pre = "features.%d.%d." % (fi, li)
            out.append(("block", pre, si, li, x))
            x = _oracle_block(x, O._sub(sd, pre), cfg, si, li)
        fi += 1
        if si < len(cfg["depths"]) - 1:
            pre = "features.%d." % fi
            out.append(("merge", pre, si, 0, x))
            x = O.swin_patch_merging(x, O._sub(sd, pre), cfg["wbits"], cfg["abits"])
            fi += 1
    return out


def test_every_block_and_merging_layer_of_swin_t_teacher_forced():
    model, img = _build_swin()
    model.train()
    cfg = SWIN_T
    sd = {k: v.detach().cpu() for k, v in model.state_dict().items()}
    with torch.no_grad():
        layers = _oracle_layers(img.cpu(), sd, cfg)
    assert sum(k == "block" for k, *_ in layers) == 12 and sum(k == "merge" for k, *_ in layers) == 3
    mods = dict(model.named_modules())
    gen = torch.Generator().manual_seed(7)
    clean, rows, explained = 0, [], []
    for kind, pre, si, li, xin in layers:
        hip = mods[pre[:-1]]
        p = {k: v.clone().requires_grad_(v.dtype.is_floating_point and "clip_val" not in k) for k, v in O._sub(sd, pre).items()}
        xo = xin.clone().requires_grad_(True)
        yo = _oracle_block(xo, p, cfg, si, li) if kind == "block" else O.swin_patch_merging(xo, p, cfg["wbits"], cfg["abits"])
        up = torch.randn(yo.shape, generator=gen)
        (yo * up).sum().backward()
        hip.zero_grad(set_to_none=True)
        xh = xin.cuda().requires_grad_(True)
        yh = hip((xh, None))
        yh = yh[0] if isinstance(yh, tuple) else yh
        (yh * up.cuda()).sum().backward()
        flips, l2_rest = _token_flips(yh, yo)
        ntok = yo.numel() // yo.shape[-1]
        assert flips <= max(ntok // 10, 49) and l2_rest < TOL and _l2(yh, yo) < 2e-2, (pre, flips, l2_rest, _l2(yh, yo))
        errs = {"dx": _l2(xh.grad, xo.grad)}
        big = max(float(v.grad.abs().max()) for k, v in p.items() if v.grad is not None)
        for n, q in hip.named_parameters():
            if q.grad is None:
                continue
            ref = p[n].grad
            assert ref is not None, (pre, n)
            if "move_" in n and float(ref.abs().max()) < 1e-4 * big:
                continue                      # offsets whose gradient is identically zero in exact arithmetic: noise
            errs[n] = _l2(q.grad, ref)
        ok = flips == 0 and max(errs.values()) < TOL
        clean += int(ok and kind == "block")
        rows.append((pre, flips, l2_rest, _l2(yh, yo), max(errs.values()), max(errs, key=errs.get)))
        if flips == 0 and not ok and kind == "block":
            # A flip-free block with a gradient off by more than 1e-3 (round 4 left two of them unexplained: 5e-3 / 6e-3 on a
            # step size).  The same block through the oracle in FP64: if the fp32 oracle is as far from fp64 as the HIP path
            # is, the deviation is the fp32 rounding of an ill-conditioned sum -- d(step) = sum_i g_i (q_i - v_i) over 10^5-10^6
            # elements of either sign, whose total is orders of magnitude below sum_i |g_i (q_i - v_i)| -- and no
            # implementation of the reference in fp32 pins it.  
            worst = max(errs, key=errs.get)
            p64 = {k: (v.detach().double().requires_grad_(v.requires_grad) if v.dtype.is_floating_point else v) for k, v in p.items()}
            x64 = xin.double().requires_grad_(True)
            (_oracle_block(x64, p64, cfg, si, li) * up.double()).sum().backward()
            g64 = p64[worst].grad
            e_hip, e_o32 = _l2(dict(hip.named_parameters())[worst].grad, g64), _l2(p[worst].grad, g64)
            explained.append((pre, worst, errs[worst], e_hip, e_o32))
            assert e_hip <= 3.0 * e_o32 + TOL, (pre, worst, errs[worst], e_hip, e_o32)
        if kind == "merge":                   # LayerNorm + one quantised linear layer: no attention to spread a flip
            assert flips <= 8 and _l2(yh, yo) < 5e-3, rows[-1]
    print("\nSwin-T W3A3 QKR: layer, tokens touched by a flip, l2(y) on the rest, l2(y) overall, worst gradient l2 (which)")
    for r in rows:
        print("   %-16s %4d  %.2e  %.2e  %.2e  %s" % r)
    for r in explained:
        print("   flip-free, gradient > 1e-3: %s %s  HIP vs fp32 oracle %.2e | HIP vs fp64 oracle %.2e | fp32 oracle vs fp64 oracle %.2e" % r)
    assert clean >= 6, rows


def test_fused_swin_block_equals_the_unfused_block():
    """Round 6: SwinTransformerBlock.forward_fused folds the cyclic shift + window partition / reverse into the LayerNorm + quantiser
    passes around the attention (ofq_layernorm_lsq_fwd_perm / _bwd_perm) and gives the MLP and patch merging the LayerNorm + quantiser
    and GELU-epilogue fusions.  Against the same block's un-fused forward() (permute kernels, separate LayerNorm and quantiser
    launches -- the path the teacher-forced oracle test above runs) on the same input, real dimensions, shifted and un-shifted,
    first and later stages, with and without a pending residual: identical integer decisions => the output is the same bits or
    differs by rounding only (< 1e-6), every gradient agrees to 1e-5 (the column sums add the same rows in another order)."""
    model, img = _build_swin()
    model.train()
    gen = torch.Generator(device="cuda").manual_seed(3)
    feats = model.features
    cases = [(feats[1][0], (2, 56, 56, 96)), (feats[1][1], (2, 56, 56, 96)), (feats[3][1], (2, 28, 28, 192)), (feats[5][2], (2, 14, 14, 384)),
             (feats[7][1], (2, 7, 7, 768))]
    for blk, shp in cases:
        x = torch.randn(shp, device="cuda", generator=gen)
        up = torch.randn(shp, device="cuda", generator=gen)
        for with_pending in (False, True):
            pend = torch.randn(shp, device="cuda", generator=gen) * 0.1 if with_pending else None
            res = {}
            for mode in ("plain", "fused"):
                blk.zero_grad(set_to_none=True)
                xi = x.clone().requires_grad_(True)
                pi = pend.clone().requires_grad_(True) if pend is not None else None
                if mode == "plain":
                    y = blk(((xi + pi) if pi is not None else xi, None))[0]
                else:
                    assert blk.attn.fused_window_plan(xi) is not None            # the path under test is the one taken
                    x2, _, m = blk.forward_fused(xi, pi)
                    y = x2 + m
                (y * up).sum().backward()
                res[mode] = (y.detach(), xi.grad.clone(), None if pi is None else pi.grad.clone(),
                             {n: q.grad.clone() for n, q in blk.named_parameters() if q.grad is not None})
            a, b = res["plain"], res["fused"]
            assert _l2(b[0], a[0]) < 1e-6, (shp, with_pending, _l2(b[0], a[0]))
            assert _l2(b[1], a[1]) < 1e-5 and (a[2] is None or _l2(b[2], a[2]) < 1e-5)
            assert a[3].keys() == b[3].keys()
            big = max(float(v.abs().max()) for v in a[3].values())
            for n in a[3]:
                if float(a[3][n].abs().max()) < 1e-6 * big:
                    continue
                assert _l2(b[3][n], a[3][n]) < 3e-5, (shp, with_pending, n, _l2(b[3][n], a[3][n]))


@pytest.mark.parametrize("model_type", ["deit", "swin"])
def test_cga_masks_and_frozen_weights_at_full_size(model_type):
    """Config C5 at its real size: CGAHooks + FusedAdamW on every quantised weight matrix of DeiT-S W2A2 (qk_reparam_type=1:
    48 tensors) / Swin-T W3A3 (cga.py:957-964: fc1, fc2, v, proj and the PatchMerging `reduction`s), 8 images.  The masks a
    step uses are the oracle's freeze_outside_boundary_weight_idx (cga.py:450-469) of the weights the step started from, bit
    for bit on every tensor; every frozen weight leaves the step bit-identical, every other quantised weight with a
    gradient moves."""
    from ofq_amd import engine
    if model_type == "deit":
        name, bits, nexp = "deit_small_distilled_patch16_224", 2, 48
    else:
        name, bits, nexp = "swin_t", 3, 51
    br = 0.005
    torch.manual_seed(0)
    model = engine.build_student(name, bits, bits, qk_reparam=True, qk_reparam_type=1).cuda()
    g = torch.Generator(device="cuda").manual_seed(3)
    imgs = torch.randn(8, 3, 224, 224, device="cuda", generator=g)
    tgt = torch.randint(0, 1000, (8,), device="cuda", generator=g)
    soft = torch.randn(8, 1000, device="cuda", generator=g)
    engine.setup_alpha(model, imgs)
    model.train()
    opt = engine.make_optimizer(model, lr=1e-3, weight_decay=0.05)
    hooks = engine.CGAHooks(model, bits, br, qk_reparam=True, model_type=model_type)
    assert len(hooks.mods) == nexp, len(hooks.mods)
    if model_type == "swin":
        assert sum(k.endswith("reduction") for k, _ in hooks.mods) == 3
    engine.train_step(model, opt, imgs, tgt, soft, cga=hooks)         # creates the AdamW state; masks of the initial weights
    before = {k: m.weight.detach().clone() for k, m in hooks.mods}
    want = {k: O.cga_freeze_idx(w.cpu(), bits, br) for k, w in before.items()}
    seen = {}
    real = hooks.after_step

    def spy(optimizer=None):
        seen.update({k: hooks.state[k][0].detach().clone() for k, _ in hooks.mods})
        return real(optimizer)
    hooks.after_step = spy
    engine.train_step(model, opt, imgs, tgt, soft, cga=hooks)
    torch.cuda.synchronize()
    frozen_total = 0
    from ofq_amd import ops
    for k, m in hooks.mods:
        frz = seen[k].float().cpu().reshape(want[k].shape)
        # (1) the CONDITIONAL statement, without exception: given the scale the device uses -- s = 2 * mean|W| with the row sum
        # formed in fp64 and rounded once (csrc/misc.hip cga_row_scale, the same arithmetic as the StatsQ kernels') -- the mask is
        # the oracle's freeze_outside_boundary_weight_idx (cga.py:450-469) bit for bit, on every element of every tensor
        s_dev = ops.statsq_fwd(before[k], bits)[1].cpu()
        assert torch.equal(frz, O.cga_freeze_idx(before[k].cpu(), bits, br, s=s_dev)), k
        # (2) against the oracle's OWN scale (torch-CPU's cascade-summed mean, which may be one ulp away from the correctly
        # rounded one): a disagreement is allowed only where that ulp decides -- the two scales of the row differ, by an ulp or two,
        # and the element's level coordinate b4 = clamp(W / s) * n - 0.5 lies within a few ulp of an edge of a +-boundaryRange
        # band under either scale (measured: 1 element of Swin-T's 27.5 M, none of DeiT-S's 21.2 M)
        diff = frz != want[k]
        if bool(diff.any()):
            W = before[k].cpu()
            s_cpu = 2 * W.abs().mean(dim=1)
            rows = diff.any(dim=1)
            ulp = torch.ldexp(torch.ones_like(s_cpu), torch.frexp(s_cpu)[1] - 24)
            assert bool(((s_dev - s_cpu).abs()[rows] > 0).all()) and bool(((s_dev - s_cpu).abs()[rows] <= 4 * ulp[rows]).all()), k
            n = float(2 ** (bits - 1))
            for sc in (s_cpu, s_dev):
                b4 = torch.clamp(W / sc[:, None], -1.0, 1.0 - 1e-6) * n - 0.5
                frac = (b4 - torch.floor(b4))[diff]                 # distance of the level coordinate from the integer below
                edge = torch.minimum((frac - (0.5 - br)).abs(), (frac - (0.5 + br)).abs())
                edge = torch.minimum(edge, torch.minimum((frac - (1.5 - br)).abs(), (frac + (0.5 - br)).abs()))
                tol = 8 * 2.0 ** -23 * (b4[diff].abs() + 1.0)
                assert bool((edge <= tol).all()), (k, float(edge.max()))
            assert int(diff.sum()) <= 2, (k, int(diff.sum()))
        f = (frz != 0).cuda()
        frozen_total += int(f.sum())
        assert torch.equal(m.weight.detach()[f], before[k][f]), k          # frozen: bit-identical across the step
        moved = (m.weight.detach()[~f] != before[k][~f]).float().mean().item()
        assert moved > 0.9, (k, moved)                                      # the others took their AdamW update
    assert frozen_total > 0
