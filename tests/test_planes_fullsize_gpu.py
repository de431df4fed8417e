"""The round-5 precision trade at the sizes the numbers are quoted on: every parameter gradient of ONE
full-size training step with the backward code GEMMs on two fp16 planes (ops.GRAD_PLANES = 2, the default) against the same
step on three bf16 planes (the fp32-EXACT products, OFQ_GRAD_PLANES=3), same weights, same batch -- plus, from the same step,
what the two-plane representation does to the gradient operands it is applied to.

Reference: the recipe runs fp32 throughout (`amp: False`, configs/ours_imagenet_recipe.attn_q.yml:27); the products in question
are autograd's of F.linear (qlinear.py:69) and of the QKR scores (attention.py:200-210).

The forward pass never touches the planes: its values and integer levels are the same bits in both runs -- asserted through the
loss (this assertion found that round 5 ran the STEM's forward GEMM on two planes; fixed, functional.CodeWeightLinearFn).  No
decision of the backward pass depends on a gradient VALUE (the LSQ masks are functions of forward values), so the two backward
passes differ by rounding only and are compared tensor by tensor.  Yardstick for "rounding only": the same THREE-plane step with
another association of the same fp32 sums (dW launched per layer instead of grouped per block: another split-K factor; v and
W_qk input gradients as two accumulating launches instead of one two-segment GEMM) -- two evaluations of the reference's own
arithmetic.  Asserted per parameter tensor, planes 2 vs planes 3 (measured values are printed; VERDICT asked for 1e-6 / 1e-5 --
1e-6 is where two EXACT evaluations already disagree, 5e-7 .. 8e-7 on the weight gradients):

    relative l2                              <= 1e-5     (measured 3e-7 .. 1.8e-6 on DeiT-S / DeiT-T, <= 5.2e-6 on Swin-T)
    max |difference| / max |gradient|        <= 1e-5     (measured <= 2.2e-6)
    step-size gradients d(s) of the LSQ quantisers: 3e-5 / 3e-5 (measured <= 1.4e-5 / 1.8e-5): sum_i g_i (q_i - v_i) over 10^4 .. 10^6 terms
        of either sign, an ill-conditioned sum in ANY fp32 evaluation (the fp32 oracle is 5e-3 from its own fp64
        run on two of them)
    attn.move_qkx_aft.bias (QKR) / attn.move_k_aft.bias (plain attention): compared on the scale of the sibling move_qkx_b4.bias /
        move_q_aft.bias -- the offset behind the qkx (k) quantiser adds a term to the scores that is constant along the softmax
        axis, so its gradient is mathematically ZERO and every evaluation returns rounding noise (1e-9 next to 1e-3)
    and every bound is widened to three times what the two exact evaluations disagree by on that tensor (Swin-T's move_qkx_b4.bias:
        5.2e-6 both ways)

and for the operands (ops.PLANE_PROBE hands every fp32 gradient operand to the test right before its GEMM splits it): the split is
EMULATED in torch -- x 2^E with the launch maximum in [2^14, 2^15), hi = fp16(x), lo = fp16(x - hi), fp16 denormals kept -- and the
representation error ||x - hi - lo|| / ||x|| and max |x - hi - lo| / max |x| asserted <= 2^-22 (measured: printed), next to the share
of NON-ZERO elements whose low plane is a denormal (|x| < 2^-17 max: absolute instead of relative precision) and the share of the
tensor's energy they carry."""
import copy
import math

import pytest
import torch

pytestmark = pytest.mark.gpu

CASES = [("deit_small_distilled_patch16_224", 2, True, 128), ("deit_tiny_distilled_patch16_224", 4, False, 256), ("swin_t", 3, True, 128)]


def _grads(engine, base, batch, planes, other_association=False):
    from ofq_amd import ops
    import ofq_amd.functional as Fn
    old = ops.GRAD_PLANES, Fn.DW_GROUP, ops.NT_CONCAT
    ops.GRAD_PLANES = planes
    if other_association:
        Fn.DW_GROUP, ops.NT_CONCAT = False, False
    try:
        model = copy.deepcopy(base).train()
        opt = engine.make_optimizer(model, lr=0.0, weight_decay=0.0)
        loss = engine.train_step(model, opt, *batch)
        torch.cuda.synchronize()
        return float(loss.detach()), {n: p.grad.detach().double().cpu() for n, p in model.named_parameters() if p.grad is not None}
    finally:
        ops.GRAD_PLANES, Fn.DW_GROUP, ops.NT_CONCAT = old


def _kind(name):
    parts = [p for p in name.split(".") if not p.isdigit()]
    return ".".join(parts[-3:])


def _is_step(name):
    return name.endswith(".s") or name.endswith("clip_val")


@pytest.mark.parametrize("cfg", CASES, ids=["deit_s_w2a2_qkr_128", "deit_t_w4a4_256", "swin_t_w3a3_qkr_128"])
def test_every_parameter_gradient_two_planes_vs_three_at_full_size(cfg):
    from ofq_amd import engine, ops
    name, bits, qkr, nimg = cfg
    torch.manual_seed(0)
    base = engine.build_student(name, bits, bits, qk_reparam=qkr).cuda()
    g = torch.Generator(device="cuda").manual_seed(11)
    batch = (torch.randn(nimg, 3, 224, 224, device="cuda", generator=g), torch.randint(0, 1000, (nimg,), device="cuda", generator=g),
             torch.randn(nimg, 1000, device="cuda", generator=g))
    engine.setup_alpha(base, batch[0][:16])

    stats = {}

    def probe(kind, A, ks, axis, ncols):
        A = A.detach()
        if ncols is not None:
            A = A[:, :ncols]
        x = A.double()
        if ks is not None:
            ks = ks.detach().double().reshape(-1)
            if axis == 1:
                x = x * ks[None, :]
            else:
                x = x * ks.repeat((x.shape[0] + ks.numel() - 1) // ks.numel())[:x.shape[0], None]
        ax = x.abs()
        amax = float(ax.max())
        if amax == 0.0 or not math.isfinite(amax):
            return
        # the kernels' scale: the power of two that puts the maximum into [2^14, 2^15)
        E = 14 - math.floor(math.log2(amax))
        xs = (x * 2.0 ** E).float()
        hi = xs.half()
        lo = (xs - hi.float()).half()
        err = xs.double() - hi.double() - lo.double()
        nz = ax > 0
        small = nz & (ax < amax * 2.0 ** -17)
        key = (kind, tuple(A.shape))
        st = stats.setdefault(key, {"n": 0, "l2": 0.0, "mx": 0.0, "share": 0.0, "energy": 0.0, "zero": 0.0})
        st["n"] += 1
        st["l2"] = max(st["l2"], float(err.norm() / xs.double().norm()))
        st["mx"] = max(st["mx"], float(err.abs().max() / xs.double().abs().max()))
        st["share"] = max(st["share"], float(small.sum()) / max(1.0, float(nz.sum())))
        st["energy"] = max(st["energy"], float((x[small] ** 2).sum() / (x ** 2).sum()))
        st["zero"] = max(st["zero"], 1.0 - float(nz.double().mean()))

    ops.PLANE_PROBE = probe
    try:
        loss2, g2 = _grads(engine, base, batch, 2)
    finally:
        ops.PLANE_PROBE = None
    loss3, g3 = _grads(engine, base, batch, 3)
    loss3b, g3b = _grads(engine, base, batch, 3, other_association=True)
    assert loss2 == loss3 == loss3b                       # the forward pass is the same bits
    assert g2.keys() == g3.keys() == g3b.keys() and len(g2) > 50

    worst = {}
    bad = []
    for n in g3:
        ref, a, b = g3[n], g2[n], g3b[n]
        den2, denm = float(ref.norm()) + 1e-300, float(ref.abs().max()) + 1e-300
        if n.endswith("move_qkx_aft.bias") or n.endswith("move_k_aft.bias"):     # mathematically zero (see the module text):
            sib = g3[n.replace("move_qkx_aft", "move_qkx_b4").replace("move_k_aft", "move_q_aft")]     # the sibling's scale
            den2, denm = float(sib.norm()) + 1e-300, float(sib.abs().max()) + 1e-300
        l2, mx = float((a - ref).norm()) / den2, float((a - ref).abs().max()) / denm
        l2e, mxe = float((b - ref).norm()) / den2, float((b - ref).abs().max()) / denm
        k = _kind(n)
        w = worst.setdefault(k, [0.0, 0.0, 0.0, 0.0])
        w[0], w[1], w[2], w[3] = max(w[0], l2), max(w[1], mx), max(w[2], l2e), max(w[3], mxe)
        lim2, limm = (3e-5, 3e-5) if _is_step(n) else (1e-5, 1e-5)
        lim2, limm = max(lim2, 3.0 * l2e), max(limm, 3.0 * mxe)      # ... or three times what two EXACT evaluations disagree by
        if not (l2 <= lim2 and mx <= limm):
            bad.append((n, l2, mx))
    print("\n%s: parameter gradients, two fp16 planes vs three bf16 planes (exact); yardstick = exact vs exact in another association" % name)
    print("%-44s %12s %12s %14s %14s" % ("parameter kind (worst over layers)", "rel l2", "max/max", "exact: rel l2", "exact: max/max"))
    for k, w in sorted(worst.items(), key=lambda kv: -kv[1][0]):
        print("%-44s %12.2e %12.2e %14.2e %14.2e" % (k, *w))
    print("gradient operands of the two-plane GEMMs, the split emulated in torch (worst over the launches of a kind):")
    print("%-8s %-18s %8s %14s %14s %24s %20s %14s" % ("family", "operand shape", "launches", "repr. rel l2", "repr. max/max",
                                                       "non-zero below 2^-17 max", "their energy share", "exact zeros"))
    for (kind, shape), st in sorted(stats.items()):
        print("%-8s %-18s %8d %14.3e %14.3e %24.3e %20.3e %14.3e" % (kind, "x".join(map(str, shape)), st["n"], st["l2"], st["mx"],
                                                                   st["share"], st["energy"], st["zero"]))
    assert not bad, bad[:8]
    assert stats, "no two-plane GEMM was probed"
    assert max(st["l2"] for st in stats.values()) <= 2.0 ** -22 and max(st["mx"] for st in stats.values()) <= 2.0 ** -22
    print("worst rel l2 over all parameters: planes 2 vs 3 %.2e (step sizes included); exact vs exact %.2e"
          % (max(w[0] for w in worst.values()), max(w[2] for w in worst.values())))
