"""The two-plane fp16 form of the backward code GEMMs (round 5; include/ofq_hip.h: ofq_qgemm_bf16s_nt with `amax`, csrc/qgemm_planes.hip
split2_f16) against fp64 and against the three-plane bf16 form (the exact fp32 product of rounds 1-4).  Reference ops: autograd of
F.linear (qlinear.py:69) and of the QKR scores (attention.py:207-210).

What "fp32-grade on the scale of the tensor" means, as tested here: the fp32 operand dY * scale is multiplied by the power of
two that puts the launch's largest magnitude into [2^14, 2^15) and split into hi = rne_f16(x), lo = rne_f16(x - hi); elements
within 2^-17 of the maximum keep 2^-24 relative precision (fp32's own), smaller ones an ABSOLUTE error of 2^-39 of the maximum.
Bounds below: component-wise 1e-6 of sum_k |a_k b_k| for rows within 1e-4 of the largest row (the three-plane kernels are held
to 2e-7 .. 1e-6 there: fp32 accumulation dominates both), and an absolute bound 1e-9 x the tensor maximum everywhere."""
import numpy as np
import pytest
import torch

from detgen import det_normalish, det_uniform
from util import T, rel_err

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ops():
    from ofq_amd import ops as o
    return o


def _operands(M, N, K, seed, row_lo=1e-4):
    rs = np.random.RandomState(seed)
    dy = (T(det_normalish((M, K), seed, 1.0)) * T(det_uniform((M, 1), seed + 1, row_lo, 10.0))).cuda()
    ks = T(det_uniform((K,), seed + 2, 0.01, 0.1)).cuda()
    wcodes = torch.from_numpy((2 * rs.randint(-8, 8, (K, N)) + 1).astype(np.int8)).cuda()
    return dy, ks, wcodes


def _check(out, dy, ks, wcodes, alpha):
    a = dy.double() * ks.double()
    ref = alpha * (a @ wcodes.double())
    den = alpha * (a.abs() @ wcodes.double().abs()) + 1e-300
    err = (out.double() - ref).abs()
    rowmax = a.abs().amax(1)
    big = rowmax >= 1e-4 * rowmax.max()
    assert float((err[big] / den[big]).max()) < 1e-6                       # fp32-grade where fp32 itself is
    kw = float(wcodes.double().abs().sum(0).max())
    assert float(err.max()) < 1e-9 * float(a.abs().max()) * kw * alpha + 1e-30 or float((err / den).max()) < 1e-6


@pytest.mark.parametrize("mnk", [(1024, 384, 384), (792, 1536, 384), (640, 384, 2304), (515, 400, 128), (300, 96, 192), (130, 1100, 64)])
def test_two_plane_dx_gemm_is_fp32_grade(ops, mnk):
    """ofq_qgemm_bf16s_nt / _nt_sk with fp16 codes: classic, streaming (whole and cut tiles) and narrow kernels vs fp64; whole-tile
    streaming launches equal the classic kernel bit for bit; launch-to-launch identical; accumulate form."""
    M, N, K = mnk
    dy, ks, wcodes = _operands(M, N, K, 31)
    wT = ops.codes_transpose_f16(wcodes)
    assert wT.dtype == torch.float16 and torch.equal(wT.float(), wcodes.float().t())
    classic = ops.qgemm_bf16s_nt(dy, wT, ks, 0.25, sk=False)
    _check(classic, dy, ks, wcodes, 0.25)
    exact = ops.qgemm_bf16s_nt(dy, ops.codes_transpose_bf16(wcodes), ks, 0.25, sk=False)       # three bf16 planes
    assert rel_err(classic.cpu(), exact.cpu()) < 3e-6          # (two fp32 accumulations over K products)
    if N > 128 and K % 64 == 0:
        tiles = ((M + 127) // 128) * ((N + (383 if N > 256 else 255)) // (384 if N > 256 else 256))
        for wgs in sorted({tiles, 1, min(7, tiles * K // 64), min(100, tiles * K // 64)}):
            out = torch.full((M, N), float("nan"), device="cuda")
            ops.qgemm_bf16s_nt_sk([(dy, wT, ks, 0.25)], out, wgs=wgs)
            _check(out, dy, ks, wcodes, 0.25)
            if tiles % wgs == 0:
                assert torch.equal(out, classic), wgs
            again = torch.full((M, N), float("nan"), device="cuda")
            ops.qgemm_bf16s_nt_sk([(dy, wT, ks, 0.25)], again, wgs=wgs)
            assert torch.equal(out, again), wgs
        assert ops.nt_sk_error(dy.device) == 0
    base = classic.clone()
    ops.qgemm_bf16s_nt(dy, wT, ks, 0.25, out=base, accumulate=True, sk=False)
    assert rel_err(base.cpu(), 2 * classic.cpu()) < 1e-6
    # without a k-scale vector
    plain = ops.qgemm_bf16s_nt(dy, wT, None, 0.5, sk=False)
    _check(plain, dy, torch.ones_like(ks), wcodes, 0.5)


def test_two_plane_scale_follows_the_maximum(ops):
    """Tensors whose magnitude is far from 1 (1e-12 .. 1e+12), an all-zero tensor, one huge outlier and a non-finite element:
    the power-of-two scale comes from the maximum word, so the relative accuracy does not depend on the magnitude; zeros stay
    zeros; an outlier costs the small rows absolute (not relative) accuracy; inf / nan propagate as they would in fp32."""
    M, N, K = 512, 384, 256
    dy, ks, wcodes = _operands(M, N, K, 7, row_lo=1e-2)
    wT = ops.codes_transpose_f16(wcodes)
    for mag in (1e-12, 1e-6, 1.0, 1e6, 1e12):
        d = (dy * mag).contiguous()
        _check(ops.qgemm_bf16s_nt(d, wT, ks, 0.25, sk=False), d, ks, wcodes, 0.25)
    z = torch.zeros_like(dy)
    assert float(ops.qgemm_bf16s_nt(z, wT, ks, 0.25, sk=False).abs().max()) == 0.0
    d = dy.clone()
    d[3, 5] = 1e9                                                       # 8 orders above everything else
    out = ops.qgemm_bf16s_nt(d, wT, ks, 0.25, sk=False)
    a = d.double() * ks.double()
    ref = 0.25 * (a @ wcodes.double())
    assert float((out.double() - ref).abs().max()) < 1e-9 * 1e9 * 0.1 * K * 15 * 0.25       # absolute, on the outlier's scale
    assert rel_err(out[3].cpu(), ref[3].float().cpu()) < 1e-6
    for bad in (float("inf"), float("nan")):
        d = dy.clone()
        d[9, 1] = bad
        out = ops.qgemm_bf16s_nt(d, wT, ks, 0.25, sk=False)
        assert not bool(torch.isfinite(out[9]).all())


@pytest.mark.parametrize("shape", [(792, 384, 384), (1188, 1536, 384), (396, 384, 1536), (2000, 2304, 384), (640, 384, 768)])
def test_two_plane_dw_gemm_matches_three_planes_and_fp64(ops, shape):
    """ofq_qgemm_bf16s_tn / _tn_group with a maximum word (the wide kernels): against fp64 and the three-plane form; the grouped
    launch equals the single launches bit for bit in either form."""
    Ktok, Mo, Nc = shape
    rs = np.random.RandomState(5)
    dy = (T(det_normalish((Ktok, Mo), 91, 1.0)) * T(det_uniform((Ktok, 1), 92, 1e-3, 10.0))).cuda()
    codes = torch.from_numpy(rs.randint(-8, 8, (Ktok, Nc)).astype(np.int8)).cuda()
    S = 198 if Ktok % 198 == 0 else Ktok
    s = T(det_uniform((S,), 93, 0.1, 1.0)).cuda()
    baft = T(det_uniform((Nc,), 94, -0.05, 0.05)).cuda()
    import ofq_oracle as O
    ae = O.lsq_effective_scale(s.cpu(), 0.01)[torch.arange(Ktok) % S].double().cuda()
    db = dy.double().sum(0)
    ref = (dy.double() * ae[:, None]).t() @ codes.double() + db[:, None] * baft.double()[None, :]
    den = ((dy.double() * ae[:, None]).abs().t() @ codes.double().abs()) + 1e-30
    res = {}
    for planes in (2, 3):
        dW, dbg = ops.qgemm_bf16s_tn(dy, codes, s, S, 0.01, None, baft, split=3, compute_db=True, planes=planes)
        assert float(((dW.double() - ref).abs() / den).max()) < 1e-6, planes
        assert rel_err(dbg.cpu(), db.float().cpu()) < 1e-5
        job = {"dy2d": dy, "xcodes2d": codes, "lsq_s": s, "S": S, "gscale": 0.01, "baft": baft,
               "dW": torch.full((Mo, Nc), float("nan"), device="cuda"), "db": torch.full((Mo,), float("nan"), device="cuda")}
        ops.qgemm_bf16s_tn_group([job], split=3, planes=planes)
        assert torch.equal(job["dW"], dW) and torch.equal(job["db"], dbg), planes
        res[planes] = dW
    assert rel_err(res[2].cpu(), res[3].cpu()) < 1e-6


@pytest.mark.parametrize("S", [1, 7, 14, 28, 31, 32, 33, 49])
def test_wide_dw_gemm_with_a_step_vector_shorter_than_a_k_tile(ops, S):
    """Swin's 4-D MLP quantisers carry S = 7 / 14 / 28 token steps (one per row of the feature map); until round 6 such a layer's dW
    ran on the narrow three-plane kernel whatever its width (the wide kernels' incremental `k mod S` assumed S >= 32).  Wide
    shapes with short step vectors: against fp64 in both plane forms, single == grouped bit for bit, and every split."""
    rs = np.random.RandomState(S)
    import ofq_oracle as O
    for Ktok, Mo, Nc in ((1568, 384, 1536), (1568 + 40, 1536, 384), (784, 192, 768), (1000, 768, 192 + 64)):
        dy = (T(det_normalish((Ktok, Mo), 91 + S, 1.0)) * T(det_uniform((Ktok, 1), 92, 1e-3, 10.0))).cuda()
        codes = torch.from_numpy(rs.randint(-4, 4, (Ktok, Nc)).astype(np.int8)).cuda()
        s = T(det_uniform((S,), 93 + S, 0.1, 1.0)).cuda()
        baft = T(det_uniform((Nc,), 94, -0.05, 0.05)).cuda()
        ae = O.lsq_effective_scale(s.cpu(), 0.01)[torch.arange(Ktok) % S].double().cuda()
        db = dy.double().sum(0)
        ref = (dy.double() * ae[:, None]).t() @ codes.double() + db[:, None] * baft.double()[None, :]
        den = ((dy.double() * ae[:, None]).abs().t() @ codes.double().abs()) + 1e-30
        assert ops.tn_groupable(Ktok, Mo, Nc, S, dy.stride(0), codes.stride(0)) == (Nc >= ops.TN_GROUP_MIN_N)
        for planes in (2, 3):
            for split in (1, 3, None):
                dW, dbg = ops.qgemm_bf16s_tn(dy, codes, s, S, 0.01, None, baft, split=split, compute_db=True, planes=planes)
                assert float(((dW.double() - ref).abs() / den).max()) < 1e-6, (planes, split, Ktok)
                assert rel_err(dbg.cpu(), db.float().cpu()) < 1e-5
            dW, dbg = ops.qgemm_bf16s_tn(dy, codes, s, S, 0.01, None, baft, split=3, compute_db=True, planes=planes)
            job = {"dy2d": dy, "xcodes2d": codes, "lsq_s": s, "S": S, "gscale": 0.01, "baft": baft,
                   "dW": torch.full((Mo, Nc), float("nan"), device="cuda"), "db": torch.full((Mo,), float("nan"), device="cuda")}
            ops.qgemm_bf16s_tn_group([job], split=3, planes=planes)
            assert torch.equal(job["dW"], dW) and torch.equal(job["db"], dbg), planes


def test_two_plane_attention_backward_matches_three_planes(ops):
    """dqkx (streaming, stacked heads) and dxq (128 x 384 tiles) at the DeiT-S geometry: two planes vs three and vs fp64; the pad
    columns of dS hold NaN on purpose (never read: the maximum word covers the real columns only)."""
    B, H, N, C = 4, 6, 198, 384
    Np = 208
    rs = np.random.RandomState(3)
    dS = torch.full((B, H, N, Np), float("nan"), device="cuda")
    dS[..., :N] = T(rs.randn(B, H, N, N).astype(np.float32) * 1e-3).cuda() * T(det_uniform((B, H, N, 1), 5, 1e-3, 1.0)).cuda()
    xc = torch.from_numpy(rs.randint(-2, 2, (B, N, C)).astype(np.int8)).cuda()
    qc = torch.from_numpy(rs.randint(-2, 2, (B, N, H, C)).astype(np.int8)).cuda()
    sx = T(0.05 + rs.rand(N).astype(np.float32)).cuda()
    sq = T(0.05 + rs.rand(N * H).astype(np.float32)).cuda()
    bax = T(rs.rand(C).astype(np.float32) * 0.1).cuda()
    import ofq_oracle as O
    ax = O.lsq_effective_scale(sx.cpu(), 0.01).double().cuda()
    aq = O.lsq_effective_scale(sq.cpu(), 0.013).view(N, H).double().cuda()
    xh = ax[None, :, None] * xc.double() + bax.double()
    ref_q = torch.einsum("bhnm,bnc->bmhc", dS[..., :N].double(), xh)
    ref_x = torch.einsum("bhnm,bmhc->bnc", dS[..., :N].double(), aq[None, :, :, None] * qc.double())
    out = {}
    for planes in (2, 3):
        dq = ops.qattn_dqkx(dS, xc, sx, 0.01, bax, B, H, N, C, Np, planes=planes)
        dx = ops.qattn_dxq(dS, qc, sq, 0.013, B, H, N, C, Np, planes=planes)
        assert rel_err(dq.cpu(), ref_q.float().cpu()) < 2e-6 and rel_err(dx.cpu(), ref_x.float().cpu()) < 2e-6, planes
        out[planes] = (dq, dx)
    assert rel_err(out[2][0].cpu(), out[3][0].cpu()) < 1e-6 and rel_err(out[2][1].cpu(), out[3][1].cpu()) < 1e-6


def _word(w):
    return float(w.view(torch.float32).max())        # (the slots of the group; the words between them stay zero)


def test_absmax_kernel_and_producer_by_products(ops):
    """ofq_absmax_f32 and the amax_out by-products of the gradient-producing backward kernels (ofq_lsq_bwd, ofq_layernorm_bwd /
    _lsq_bwd, ofq_softmax_lsq_bwd, ofq_qattn_dp_softmax_bwd, ofq_qgemm_i8_lsq_bwd): the word group's maximum is exactly
    max |tensor| (a maximum does not round), NaN is reported, and a strided view works."""
    g = torch.Generator(device="cuda").manual_seed(2)
    x = torch.randn(1000, 384, device="cuda", generator=g) * torch.logspace(-6, 2, 1000, device="cuda").unsqueeze(1)
    assert _word(ops.absmax(x)) == float(x.abs().max())
    assert _word(ops.absmax(x[:, 64:192])) == float(x[:, 64:192].abs().max())             # strided rows
    assert _word(ops.absmax(x[:, :198])) == float(x[:, :198].abs().max())                 # cols % 4 != 0: the stock reduction
    xn = x.clone()
    xn[17, 3] = float("nan")
    assert np.isnan(_word(ops.absmax(xn)))
    if ops.GRAD_PLANES != 2:
        pytest.skip("the producers raise a maximum word in two-plane mode only (run under OFQ_GRAD_PLANES=3)")
    # LayerNorm backward (plain)
    R, C = 792, 384
    xx = torch.randn(R, C, device="cuda", generator=g)
    gamma, beta = torch.rand(C, device="cuda", generator=g) + 0.5, torch.randn(C, device="cuda", generator=g)
    y, _, mean, rstd = ops.layernorm_fwd(xx, gamma, beta, 1e-6)
    dy = torch.randn(R, C, device="cuda", generator=g) * 1e-3
    dx, _, _ = ops.layernorm_bwd(dy, xx, mean, rstd, gamma)
    assert ops.amax_of(dx) is not None and _word(ops.amax_of(dx)) == float(dx.abs().max())
    assert ops.amax_of(dx.view(4, 198, C).reshape(-1, C)) is ops.amax_of(dx)            # a reshape in between keeps the word
    # LSQ backward, per token and per channel
    s = torch.rand(198, device="cuda", generator=g) * 0.1 + 0.02
    geom = ops.LsqGeom(4, 198, C, C, 0, -2, 1, 4 * C)
    dxl, _, _, _ = ops.lsq_bwd(dy, xx, s, torch.zeros(C, device="cuda"), geom)
    assert _word(ops.amax_of(dxl)) == float(dxl.abs().max())
    # three launches writing column slices of ONE tensor share its word (plain attention: dq | dk | dv -> the qkv projection)
    wide = torch.full((R, 3 * C), float("nan"), device="cuda")
    am = ops.amax_out(wide.device)
    if am is not None:
        x3 = torch.randn(R, 3 * C, device="cuda", generator=g)
        for i, scale in enumerate((1e-3, 7.0, 1e-5)):
            ops.lsq_bwd(dy * scale, x3[:, i * C:], s, torch.zeros(C, device="cuda"),
                        ops.LsqGeom(4, 198, C, C, 0, -2, 1, 4 * C, ldx=3 * C, ldy=C), dx=wide[:, i * C:], amax_word=am)
        assert not bool(torch.isnan(wide).any())
        assert _word(am) == float(wide.abs().max()) and ops.amax_of(wide[:, C:]) is None     # (the slices are not tagged)
    # softmax-LSQ backward (in place) and the fused dP + softmax backward
    B, H, N, d, Np = 2, 3, 198, 64, 208
    prob = torch.softmax(torch.randn(B, H, N, Np, device="cuda", generator=g), -1)
    gs = torch.randn(B, H, N, Np, device="cuda", generator=g) * 1e-2
    sm = torch.rand(N, device="cuda", generator=g) * 0.05 + 0.01
    dS, _ = ops.softmax_lsq_bwd(gs.clone(), prob, sm, B * H * N, N, Np, N, 0.125, 3, B * H * N, inplace=True)
    assert _word(ops.amax_of(dS)) == float(dS[..., :N].abs().max())
    dO = torch.randn(B, N, H * d, device="cuda", generator=g) * 1e-2
    vc = torch.randint(-2, 2, (B, N, H * d), dtype=torch.int8, device="cuda", generator=g)
    sv = torch.rand(H * d, device="cuda", generator=g) * 0.1 + 0.05
    bav = torch.rand(H * d, device="cuda", generator=g) * 0.01
    dS2, _, _ = ops.qattn_dp_softmax_bwd(dO, vc, sv, 0.01, bav, prob, sm, 0.125, 3, B, H, N, d, Np)
    assert _word(ops.amax_of(dS2)) == float(dS2[..., :N].abs().max())
