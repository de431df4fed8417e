"""GPU parity tests, kernel level: every C-ABI entry point against the oracle / the reference goldens.
Integer levels and everything that is pure elementwise fp32 are compared BIT-EXACTLY; reductions and GEMMs
within 1e-5 (the tier's bound is 1e-3).  Run on the MI355X box:  pytest -m gpu"""
import math

import numpy as np
import pytest
import torch

import ofq_oracle as O
from detgen import det_uniform, det_normalish
from util import load_golden, group, case_names, T, rel_err

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ops():
    assert torch.cuda.is_available(), "the -m gpu tests need a HIP device"
    from ofq_amd import ops as _ops
    _ops.lib()
    return _ops


def G(a):
    return torch.as_tensor(np.ascontiguousarray(a)).cuda()


# ------------------------------------------------------------------------------------------------ StatsQ
def test_statsq_golden_bit_exact_given_scale(ops):
    d = load_golden("g1_statsq")
    for c in range(int(d["ncases"])):
        g = group(d, "c%d" % c)
        bits = int(g["shape"][2])
        out, s, lv = ops.statsq_fwd(G(g["W"]), bits, want_levels=True, scale=G(g["s"]))
        assert torch.equal(lv.cpu(), T(g["L"])), c
        assert torch.equal(out.cpu(), T(g["y"])), c


def test_statsq_golden_own_scale(ops):
    d = load_golden("g1_statsq")
    nbad = ntot = 0
    for c in range(int(d["ncases"])):
        g = group(d, "c%d" % c)
        bits = int(g["shape"][2])
        out, s, lv = ops.statsq_fwd(G(g["W"]), bits, want_levels=True)
        assert rel_err(s.cpu(), g["s"]) < 5e-7
        # a level may only differ where c*n-0.5 sits within an ulp of a rounding tie (the kernel sums |W| in fp64 and rounds
        # once, torch-CPU in an fp32 cascade: the scales can differ by one ulp)
        bad = (lv.cpu() != T(g["L"]))
        nbad += int(bad.sum())
        ntot += bad.numel()
        assert int(bad.sum()) <= 1
        assert rel_err(out.cpu(), g["y"]) < 1e-6 or bad.any()
    assert nbad <= 1e-5 * ntot, (nbad, ntot)          # measured: 0 of 127 k levels (full-size tensors: ~4e-7)


def test_statsq_real_shapes_vs_oracle(ops):
    for (r, c, bits, seed) in [(384, 384, 2, 1), (1536, 384, 2, 2), (384, 1536, 3, 3), (2304, 384, 2, 4), (576, 192, 4, 5)]:
        W = T(det_normalish((r, c), seed, 0.02))
        y, L, s = O.statsq(W, bits)
        out, sg, lv = ops.statsq_fwd(W.cuda(), bits, want_levels=True)
        assert rel_err(sg.cpu(), s.squeeze()) < 5e-7
        mism = (lv.cpu().int() != L).float().mean().item()
        assert mism < 1e-5, (r, c, mism)
        out2, _, lv2 = ops.statsq_fwd(W.cuda(), bits, want_levels=True, scale=s.squeeze().cuda())
        assert torch.equal(lv2.cpu().int(), L)
        assert torch.equal(out2.cpu(), y.detach())


def test_statsq_levels_given_the_devices_scale_and_where_the_own_scales_may_differ(ops):
    """The airtight form of the documented limit of "bit-exact integer levels" (statsq.py:138-147).
    (1) CONDITIONAL: with the scale the device computed (fp64 row sum, rounded once) injected into the oracle's formula, levels
    and values are the oracle's bit for bit -- every element, every case, no exception.
    (2) With the oracle's OWN scale (torch-CPU's cascade-summed mean) a level may differ only where the two scales of that row
    differ (by an ulp or two) and the element's coordinate c * n - 0.5 lies within a few ulp of a rounding tie under either scale."""
    cases = []
    d = load_golden("g1_statsq")
    for c in range(int(d["ncases"])):
        g = group(d, "c%d" % c)
        cases.append((T(g["W"]), int(g["shape"][2])))
    for (r, c, bits, seed) in [(384, 384, 2, 1), (1536, 384, 2, 2), (384, 1536, 3, 3), (2304, 384, 2, 4), (576, 192, 4, 5), (768, 3072, 3, 6)]:
        cases.append((T(det_normalish((r, c), seed, 0.02)), bits))
    ndiff = 0
    for W, bits in cases:
        out, s_dev, lv = ops.statsq_fwd(W.cuda(), bits, want_levels=True)
        s_dev = s_dev.cpu()
        y_c, L_c, _ = O.statsq(W, bits, s=s_dev)
        assert torch.equal(lv.cpu().int(), L_c) and torch.equal(out.cpu(), y_c.detach())
        y_o, L_o, s_o = O.statsq(W, bits)
        s_o = s_o.squeeze(1)
        bad = lv.cpu().int() != L_o
        ndiff += int(bad.sum())
        ulp = torch.ldexp(torch.ones_like(s_o), torch.frexp(s_o)[1] - 24)
        assert bool(((s_dev - s_o).abs() <= 4 * ulp).all())                  # the two scales: a few ulp apart at most (cascade sum), every row
        if bool(bad.any()):
            rows = bad.any(dim=1)
            assert bool(((s_dev - s_o).abs()[rows] > 0).all())
            n = float(2 ** (bits - 1))
            for sc in (s_o, s_dev):
                t = torch.clamp(W / sc[:, None], -1.0, 1.0 - 1e-6) * n - 0.5
                tie = ((t - torch.floor(t)) - 0.5).abs()[bad]
                assert bool((tie <= 8 * 2.0 ** -23 * (t[bad].abs() + 1.0)).all()), float(tie.max())
    print("levels that differ under the oracle's own scale: %d" % ndiff)


# ------------------------------------------------------------------------------------------------ LSQ
def _golden_geom(ops, name, g):
    x = g["x"]
    lo, hi = int(g["lohi"][0]), int(g["lohi"][1])
    ns = g["s"].size
    M = x.size // ns
    if name.startswith("token"):
        S, inner = x.shape[-2], x.shape[-1]
        return ops.LsqGeom(x.size // (S * inner), S, inner, 0, 0, lo, hi, M)
    if name.startswith("chan"):
        return ops.LsqGeom(x.size // x.shape[-1], 1, x.shape[-1], 0, 1, lo, hi, M)
    if name.startswith("img"):
        return ops.LsqGeom(x.shape[0], x.shape[1], x.shape[2] * x.shape[3], 0, 0, lo, hi, M)
    if name == "convw":
        return ops.LsqGeom(1, x.shape[0], x.size // x.shape[0], 0, 0, lo, hi, M)
    if name == "roww":
        return ops.LsqGeom(1, x.shape[0], x.shape[1], 0, 0, lo, hi, M)
    if name == "tensor":
        return ops.LsqGeom(x.shape[0], 1, x.shape[1], 0, 0, lo, hi, M)
    raise AssertionError(name)


def test_lsq_golden_all_variants_bit_exact(ops):
    d = load_golden("g2_lsq")
    n = 0
    for nme in case_names(d):
        g = group(d, nme)
        geom = _golden_geom(ops, nme, g)
        x, s, gy = G(g["x"]), G(g["s"]).reshape(-1), G(g["g"])
        y, codes = ops.lsq_fwd(x, s, None, None, geom, want_codes=True)
        assert torch.equal(y.cpu().reshape(g["y"].shape), T(g["y"])), nme
        dx, ds, _, _ = ops.lsq_bwd(gy, x, s, None, geom)
        assert torch.equal(dx.cpu().reshape(g["dx"].shape), T(g["dx"])), nme
        assert rel_err(ds.cpu().reshape(-1), g["ds"].reshape(-1)) < 1e-5, nme
        # integer levels == oracle levels, and inside [lo, hi]
        lo, hi = geom.lo, geom.hi
        cview = codes.view(torch.uint8) if lo == 0 else codes       # unsigned ranges are stored as uint8
        assert int(cview.min()) >= lo and int(cview.max()) <= hi
        n += 1
    assert n >= 25


def _oracle_lsq_sandwich(x, s, b4, baft, mode, bits, unsigned, gelu, H=1):
    xx = torch.nn.functional.gelu(x) if gelu else x
    xx = xx + b4
    if mode == "token":
        B, N, Cw = xx.shape
        y = O.lsq_token(xx.reshape(B, N * H, Cw // H), s, bits, unsigned).reshape(B, N, Cw)
    else:
        y = O.lsq_channel(xx, s, bits, unsigned)
    return y + baft


@pytest.mark.parametrize("case", [
    dict(B=4, N=198, C=384, H=1, mode="token", bits=2, unsigned=False, gelu=False),
    dict(B=3, N=198, C=1536, H=1, mode="token", bits=2, unsigned=True, gelu=True),
    dict(B=2, N=198, C=2304, H=6, mode="token", bits=2, unsigned=False, gelu=False),     # qkx: s per (token, head)
    dict(B=4, N=198, C=384, H=1, mode="channel", bits=2, unsigned=False, gelu=False),
    dict(B=5, N=198, C=192, H=1, mode="token", bits=4, unsigned=False, gelu=False),
    dict(B=2, N=198, C=576, H=3, mode="token", bits=4, unsigned=False, gelu=False),
    dict(B=2, N=198, C=768, H=1, mode="token", bits=3, unsigned=True, gelu=True),
    dict(B=3, N=50, C=24, H=1, mode="channel", bits=3, unsigned=False, gelu=False),
    dict(B=3, N=49, C=4608, H=12, mode="token", bits=3, unsigned=False, gelu=False),     # Swin stage 3 qkx: 12 phases > 8 row-groups
    dict(B=2, N=49, C=18432, H=24, mode="token", bits=3, unsigned=False, gelu=False),    # Swin stage 4 qkx
    dict(B=2304, N=49, C=96, H=1, mode="token", bits=3, unsigned=False, gelu=False),     # Swin stage 1: 49 scales over thousands of windows (deep second-stage sum)
])
def test_lsq_sandwich_vs_oracle(ops, case):
    B, N, C, H = case["B"], case["N"], case["C"], case["H"]
    bits, unsigned, gelu = case["bits"], case["unsigned"], case["gelu"]
    lo, hi = O.lsq_bounds(bits, unsigned)
    x = T(det_normalish((B, N, C), 11, 1.0))
    b4 = T(det_uniform((C,), 12, -0.05, 0.05))
    baft = T(det_uniform((C,), 13, -0.05, 0.05))
    if case["mode"] == "token":
        ns = N * H
        s0 = O.lsq_token_init((torch.nn.functional.gelu(x) if gelu else x).reshape(B, N * H, C // H), bits, unsigned)
        geom = ops.LsqGeom(B, N * H, C // H, C, 0, lo, hi, B * C // H, prologue=int(gelu))
    else:
        ns = C
        s0 = O.lsq_channel_init(x, bits)
        geom = ops.LsqGeom(B * N, 1, C, C, 1, lo, hi, B * N, prologue=int(gelu))
    s = (s0 * T(det_uniform((ns,), 14, 0.7, 1.1))).contiguous()
    xr = x.clone().requires_grad_(True)
    sr, b4r, baftr = s.clone().requires_grad_(True), b4.clone().requires_grad_(True), baft.clone().requires_grad_(True)
    y = _oracle_lsq_sandwich(xr, sr, b4r, baftr, case["mode"], bits, unsigned, gelu, H)
    gy = T(det_uniform((B, N, C), 15, -1.0, 1.0))
    (y * gy).sum().backward()

    yg, codes = ops.lsq_fwd(x.cuda(), s.cuda(), b4.cuda(), baft.cuda(), geom, want_codes=True)
    dx, ds, db4, dbaft = ops.lsq_bwd(gy.cuda(), x.cuda(), s.cuda(), b4.cuda(), geom)
    yg, dx = yg.cpu().reshape(B, N, C), dx.cpu().reshape(B, N, C)
    if not gelu:
        assert torch.equal(yg, y.detach())
        assert torch.equal(dx, xr.grad)
    else:
        # erf differs in the last ulp between libm and the device: levels may flip on exact ties only
        flips = (yg != y.detach()).float().mean().item()
        assert flips < 1e-4
        assert rel_err(dx, xr.grad) < 1e-3 or flips > 0
    assert rel_err(ds.cpu(), sr.grad) < 2e-4 if gelu else rel_err(ds.cpu(), sr.grad) < 1e-5
    assert rel_err(db4.cpu(), b4r.grad) < 1e-5 or gelu
    assert rel_err(dbaft.cpu(), baftr.grad) < 1e-5
    assert int(codes.min()) >= lo and int(codes.max()) <= hi


def test_lsq_strided_slices_of_qkv(ops):
    # q/k/v thirds of a [B*N, 3C] projection quantised in place (attention.py:72-81)
    B, N, C, bits = 3, 198, 192, 4
    lo, hi = O.lsq_bounds(bits, False)
    qkv = T(det_normalish((B, N, 3 * C), 21, 1.0))
    b4 = T(det_uniform((3 * C,), 22, -0.05, 0.05))
    s = O.lsq_token_init(qkv[..., :C], bits, False) * 0.9
    for part in range(2):
        xs = qkv[..., part * C:(part + 1) * C]
        ref = O.lsq_token(xs + b4[part * C:(part + 1) * C], s, bits, False)
        geom = ops.LsqGeom(B, N, C, C, 0, lo, hi, B * C, ldx=3 * C, ldy=C)
        xg = qkv.cuda()
        view = xg.view(-1)[part * C:]
        y, _ = ops.lsq_fwd(view, s.cuda(), b4[part * C:(part + 1) * C].cuda().contiguous(), None, geom)
        assert torch.equal(y.cpu().reshape(B, N, C), ref)


# ------------------------------------------------------------------------------------------------ softmax + LSQ
@pytest.mark.parametrize("shape", [(2, 6, 198, 2), (3, 3, 198, 4), (2, 2, 7, 2), (1, 2, 49, 3)])
def test_softmax_lsq_vs_oracle(ops, shape):
    B, H, N, bits = shape
    ld = (N + 3) // 4 * 4
    lo, hi = O.lsq_bounds(bits, True)
    sc = T(det_normalish((B, H, N, N), 31, 3.0))
    alpha = 0.125
    p0 = torch.softmax(sc * alpha, -1)
    s = (O.lsq_token_init(p0, bits, True) * T(det_uniform((N,), 32, 0.5, 1.0))).contiguous()
    scr = sc.clone().requires_grad_(True)
    sr = s.clone().requires_grad_(True)
    prob = torch.softmax(scr * alpha, -1)
    y = O.lsq_token(prob, sr, bits, True)
    gy = T(det_uniform((B, H, N, N), 33, -1.0, 1.0))
    (y * gy).sum().backward()

    pad = torch.zeros(B, H, N, ld)
    pad[..., :N] = sc
    probg, yg = ops.softmax_lsq_fwd(pad.cuda(), s.cuda(), B * H * N, N, ld, N, alpha, hi, B * H * N)
    assert rel_err(probg.cpu()[..., :N], prob.detach()) < 1e-6
    assert float(probg.cpu()[..., N:].abs().max() if ld > N else 0.0) == 0.0
    flips = (yg.cpu()[..., :N] != y.detach()).float().mean().item()
    assert flips < 2e-4, flips
    gpad = torch.zeros(B, H, N, ld)
    gpad[..., :N] = gy
    dsc, ds = ops.softmax_lsq_bwd(gpad.cuda(), probg, s.cuda(), B * H * N, N, ld, N, alpha, hi, B * H * N, inplace=False)
    assert rel_err(dsc.cpu()[..., :N], scr.grad) < 2e-3 if flips > 0 else rel_err(dsc.cpu()[..., :N], scr.grad) < 1e-5
    assert rel_err(ds.cpu(), sr.grad) < 2e-3 if flips > 0 else rel_err(ds.cpu(), sr.grad) < 1e-5


# ------------------------------------------------------------------------------------------------ GEMM
def _ref_mm(a, b):
    return (a.double() @ b.double()).float()


@pytest.mark.parametrize("mnk", [(256, 384, 384), (396, 1536, 384), (200, 384, 1536), (198, 198, 384), (198, 64, 198),
                                 (37, 29, 50), (130, 70, 33), (64, 2304, 384)])
@pytest.mark.parametrize("ta,tb", [(False, True), (False, False), (True, False), (True, True)])
def test_gemm_all_layouts(ops, mnk, ta, tb):
    M, N, K = mnk
    A = T(det_normalish((K, M) if ta else (M, K), 41, 1.0))
    Bm = T(det_normalish((N, K) if tb else (K, N), 42, 1.0))
    bias = T(det_uniform((N,), 43))
    ref = _ref_mm(A.t() if ta else A, Bm.t() if tb else Bm) + bias
    Cg = torch.empty(M, N, device="cuda")
    ops.gemm(A.cuda(), Bm.cuda(), Cg, M, N, K, A.shape[1], Bm.shape[1], N, transA=ta, transB=tb, bias=bias.cuda())
    assert rel_err(Cg.cpu(), ref) < 1e-5


def test_gemm_is_an_exact_fmaf_chain_on_integers(ops):
    # small-integer operands: every partial sum is exact in fp32, so the result must be exact
    M, N, K = 256, 128, 384
    A = torch.from_numpy(np.random.RandomState(0).randint(-2, 2, (M, K)).astype(np.float32))
    Bm = torch.from_numpy(np.random.RandomState(1).randint(-3, 4, (N, K)).astype(np.float32))
    Cg = torch.empty(M, N, device="cuda")
    ops.gemm(A.cuda(), Bm.cuda(), Cg, M, N, K, K, K, N, transB=True)
    assert torch.equal(Cg.cpu(), (A.double() @ Bm.double().t()).float())


def test_gemm_asymmetric_identity_detects_transposes(ops):
    M = N = K = 64
    A = torch.eye(64)
    Bm = torch.arange(64 * 64, dtype=torch.float32).reshape(64, 64)     # asymmetric
    Cg = torch.empty(M, N, device="cuda")
    ops.gemm(A.cuda(), Bm.cuda(), Cg, M, N, K, K, N, N)
    assert torch.equal(Cg.cpu(), Bm)


def test_gemm_splitk_and_linear_helpers(ops):
    M, N, K = 4 * 198, 384, 1536
    x = T(det_normalish((M, K), 51, 1.0))
    W = T(det_normalish((N, K), 52, 0.05))
    b = T(det_uniform((N,), 53))
    dy = T(det_normalish((M, N), 54, 1.0))
    y = ops.linear_fwd(x.cuda(), W.cuda(), b.cuda())
    assert rel_err(y.cpu(), _ref_mm(x, W.t()) + b) < 1e-5
    dx = ops.linear_bwd_input(dy.cuda(), W.cuda())
    assert rel_err(dx.cpu(), _ref_mm(dy, W)) < 1e-5
    dW = ops.linear_bwd_weight(dy.cuda(), x.cuda())
    assert rel_err(dW.cpu(), _ref_mm(dy.t(), x)) < 1e-5
    dW2 = torch.empty(N, K, device="cuda")
    ops.gemm(dy.cuda(), x.cuda(), dW2, N, K, M, N, K, K, transA=True, split_k=7)
    assert rel_err(dW2.cpu(), _ref_mm(dy.t(), x)) < 1e-5
    assert rel_err(ops.colsum(dy.cuda()).cpu(), dy.double().sum(0).float()) < 1e-5


def test_gemm_batched_attention_shapes(ops):
    B, H, N, C = 2, 3, 198, 96
    d = C // H
    Np = 200
    xq = T(det_normalish((B, N, C), 61, 1.0))
    qkx = T(det_normalish((B, N, H, C), 62, 1.0))
    # S[b,h,n,m] = sum_c xq[b,n,c] qkx[b,m,h,c]                      (attention.py:210)
    S = torch.zeros(B, H, N, Np, device="cuda")
    ops.gemm(xq.cuda(), qkx.cuda(), S, N, N, C, C, H * C, Np, transB=True, nb0=B, nb1=H, sA=(N * C, 0),
             sB=(N * H * C, C), sC=(H * N * Np, N * Np))
    ref = torch.einsum("bnc,bmhc->bhnm", xq.double(), qkx.double()).float()
    assert rel_err(S.cpu()[..., :N], ref) < 1e-5
    # O[b,n,h*d+j] = sum_m P[b,h,n,m] v[b,m,h*d+j]                   (attention.py:219)
    P = torch.zeros(B, H, N, Np)
    P[..., :N] = T(det_uniform((B, H, N, N), 63, 0.0, 1.0))
    v = T(det_normalish((B, N, C), 64, 1.0))
    Og = torch.empty(B, N, C, device="cuda")
    ops.gemm(P.cuda(), v.cuda(), Og, N, d, N, Np, C, C, nb0=B, nb1=H, sA=(H * N * Np, N * Np), sB=(N * C, d),
             sC=(N * C, d))
    refO = (P[..., :N].double() @ v.double().reshape(B, N, H, d).permute(0, 2, 1, 3)).transpose(1, 2).reshape(B, N, C)
    assert rel_err(Og.cpu(), refO.float()) < 1e-5
    # dxq[b,n,c] = sum_h sum_m dS[b,h,n,m] qkx[b,m,h,c]   (k-batched accumulation over heads)
    dS = torch.zeros(B, H, N, Np)
    dS[..., :N] = T(det_normalish((B, H, N, N), 65, 1.0))
    dxq = torch.empty(B, N, C, device="cuda")
    ops.gemm(dS.cuda(), qkx.cuda(), dxq, N, C, N, Np, H * C, C, nb0=B, sA=(H * N * Np, 0), sB=(N * H * C, 0),
             sC=(N * C, 0), nkb=H, sAk=N * Np, sBk=C)
    refdx = torch.einsum("bhnm,bmhc->bnc", dS[..., :N].double(), qkx.double()).float()
    assert rel_err(dxq.cpu(), refdx) < 1e-5
    # accumulate flag
    ops.gemm(dS.cuda(), qkx.cuda(), dxq, N, C, N, Np, H * C, C, nb0=B, sA=(H * N * Np, 0), sB=(N * H * C, 0),
             sC=(N * C, 0), nkb=H, sAk=N * Np, sBk=C, accumulate=True)
    assert rel_err(dxq.cpu(), 2 * refdx) < 1e-5


# ------------------------------------------------------------------------------------------------ CGA
def test_cga_golden_bit_exact(ops):
    d = load_golden("g8_cga")
    for c in range(int(d["ncases"])):
        g = group(d, "c%d" % c)
        bits, br = int(g["meta"][2]), float(g["br"])
        W = G(g["W"])
        frz = ops.cga_freeze_mask(W, bits, br)
        assert torch.equal(frz.cpu(), T(g["frz"])), c
        grad = G(g["g"]).clone()
        saved = ops.cga_mask_grad_save(grad, W, frz)
        assert torch.equal(grad.cpu(), T(g["gm"]))
        Wp = torch.nn.Parameter(W.clone())
        opt = torch.optim.AdamW([Wp], lr=1e-3, weight_decay=0.05)
        Wp.grad = grad
        opt.step()
        ops.cga_restore(Wp.data, frz, saved)
        assert rel_err(Wp.detach().cpu(), g["W_after"]) < 1e-6
        fm = T(g["frz"]) == 1
        assert torch.equal(Wp.detach().cpu()[fm], T(g["W"])[fm])      # frozen weights are restored exactly


def test_cga_multi_tensor_masks_and_fused_adamw_equal_the_golden_sequence(ops):
    """All masks in one multi-tensor call (ofq_cga_freeze_mask_multi) equal the per-tensor kernel bit for bit, and the
    AdamW kernel with the mask folded in reproduces the golden mask-grad / AdamW / restore result (cga.py:953-1013)."""
    from ofq_amd.optim import FusedAdamW
    d = load_golden("g8_cga")
    cases = [group(d, "c%d" % c) for c in range(int(d["ncases"]))]
    by_cfg = {}
    for g in cases:
        by_cfg.setdefault((int(g["meta"][2]), float(g["br"])), []).append(g)
    for (bits, br), gs in by_cfg.items():
        Ws = [G(g["W"]) for g in gs] + [torch.randn(37, 50, device="cuda") * 0.02, torch.randn(1536, 384, device="cuda") * 0.02]
        masks, _ = ops.cga_freeze_mask_multi(Ws, bits, br)
        for w, m in zip(Ws, masks):
            assert torch.equal(m, ops.cga_freeze_mask(w, bits, br))
        for g, m in zip(gs, masks):
            assert torch.equal(m.cpu(), T(g["frz"]))
            Wp = torch.nn.Parameter(G(g["W"]).clone())
            opt = FusedAdamW([Wp], lr=1e-3, weight_decay=0.05)
            Wp.grad = G(g["g"]).clone()
            opt.set_frozen(Wp, m)
            opt.step()
            assert rel_err(Wp.detach().cpu(), g["W_after"]) < 1e-6
            fm = T(g["frz"]) == 1
            assert torch.equal(Wp.detach().cpu()[fm], T(g["W"])[fm])


def test_cpu_tensors_are_rejected_loudly(ops):
    with pytest.raises(RuntimeError):
        ops.statsq_fwd(torch.zeros(4, 8), 2)


# ------------------------------------------------------------------------------------------------ exact code GEMMs
@pytest.mark.parametrize("mnk", [(396, 384, 384), (792, 1536, 384), (200, 384, 1536), (130, 70, 48), (64, 2304, 384)])
def test_qgemm_i8_forward_is_exact(ops, mnk):
    M, N, K = mnk
    rs = np.random.RandomState(3)
    qa = torch.from_numpy(rs.randint(-8, 8, (M, K)).astype(np.int8))
    qw = torch.from_numpy((2 * rs.randint(-4, 4, (N, K)) + 1).astype(np.int8))
    S = 198 if M % 198 == 0 else M
    s = T(det_uniform((S,), 71, 0.1, 1.0))
    s[1] = 3e-6                                                    # below the 1e-5 floor
    cs = T(det_uniform((N,), 72, 0.01, 0.1))
    baft = T(det_uniform((K,), 73, -0.05, 0.05))
    bias = T(det_uniform((N,), 74, -0.1, 0.1))
    gscale = 1.0 / math.sqrt(7 * 4 * K)
    r = ops.rowdot_i8(qw.cuda(), baft.cuda())
    assert rel_err(r.cpu(), (qw.double() @ baft.double()).float()) < 1e-6
    y = ops.qgemm_i8_nt(qa.cuda(), qw.cuda(), bias.cuda(), cs.cuda(), 0.125, r, s.cuda(), S, gscale)
    ae = O.lsq_effective_scale(s, gscale)[torch.arange(M) % S].double()
    ref = (0.125 * cs.double()) * (ae[:, None] * (qa.double() @ qw.double().t()) + (qw.double() @ baft.double())) + bias.double()
    assert rel_err(y.cpu(), ref.float()) < 1e-6
    # same numbers as the fp32 path on the fake-quant values: x_hat = ae*qa + baft, W_hat = cs/8 * qw
    xh = (ae[:, None] * qa.double() + baft.double()).float()
    wh = ((0.125 * cs.double())[:, None] * qw.double()).float()
    y2 = ops.linear_fwd(xh.cuda(), wh.cuda(), bias.cuda())
    assert rel_err(y.cpu(), y2.cpu()) < 1e-5


@pytest.mark.parametrize("mnk", [(396, 384, 384), (792, 384, 1536), (200, 1536, 384), (130, 72, 40), (256, 384, 2304),
                                 (333, 192, 776), (150, 768, 96), (129, 200, 24)])
@pytest.mark.parametrize("nsplit", [3, 2])
def test_qgemm_bf16_split_backward(ops, mnk, nsplit):
    M, N, K = mnk
    rs = np.random.RandomState(4)
    dy = T(det_normalish((M, K), 81, 1.0)) * T(det_uniform((M, 1), 82, 1e-4, 10.0))     # wide dynamic range
    ks = T(det_uniform((K,), 83, 0.01, 0.1))
    wcodes = torch.from_numpy((2 * rs.randint(-8, 8, (K, N)) + 1).astype(np.int8))       # [o][c] layout, o = k here
    wT = ops.codes_transpose_bf16(wcodes.cuda())                                        # [c][o] bf16
    assert torch.equal(wT.float().cpu(), wcodes.float().t())
    out = ops.qgemm_bf16s_nt(dy.cuda(), wT, ks.cuda(), 0.25, nsplit=nsplit)
    ref = 0.25 * ((dy.double() * ks.double()) @ wcodes.double())
    err = float(((out.cpu().double() - ref).abs() / (((dy.double() * ks.double()).abs() @ wcodes.double().abs()) * 0.25 + 1e-30)).max())
    assert err < (2e-7 if nsplit == 3 else 2e-5), err
    out2 = ops.qgemm_bf16s_nt(dy.cuda(), wT, ks.cuda(), 0.25, out=out.clone(), accumulate=True, nsplit=nsplit)
    assert rel_err(out2.cpu(), 2 * ref.float()) < (1e-5 if nsplit == 3 else 1e-4)


def _nt_operands(M, N, K, seed):
    rs = np.random.RandomState(seed)
    dy = (T(det_normalish((M, K), seed, 1.0)) * T(det_uniform((M, 1), seed + 1, 1e-4, 10.0))).cuda()     # wide dynamic range
    ks = T(det_uniform((K,), seed + 2, 0.01, 0.1)).cuda()
    wcodes = torch.from_numpy((2 * rs.randint(-8, 8, (K, N)) + 1).astype(np.int8)).cuda()                # [o][c], o = k here
    return dy, ks, wcodes


@pytest.mark.parametrize("mnk", [(1024, 384, 384), (792, 1536, 384), (640, 384, 2304), (1000, 192, 768), (515, 400, 128), (130, 1100, 64)])
def test_streaming_dx_gemm_equals_the_tile_per_workgroup_kernel(ops, mnk):
    """ofq_qgemm_bf16s_nt_sk (reference: autograd of F.linear, qlinear.py:69).  With a workgroup count that divides the
    tile count no tile is cut: the k order of every tile is the classic kernel's and the results are the same bits.
    With cut tiles (any other count) another association of the same fp32 sums: fp64 accuracy as good as the classic
    kernel's, and launch-to-launch bit-identical (the owner adds the partials in workgroup order)."""
    M, N, K = mnk
    dy, ks, wcodes = _nt_operands(M, N, K, 91)
    wT = ops.codes_transpose_bf16(wcodes)
    ref = 0.25 * ((dy.double() * ks.double()) @ wcodes.double())
    den = ((dy.double() * ks.double()).abs() @ wcodes.double().abs()) * 0.25 + 1e-30
    classic = ops.qgemm_bf16s_nt(dy, wT, ks, 0.25, sk=False)
    tiles = ((M + 127) // 128) * ((N + (383 if N > 256 else 255)) // (384 if N > 256 else 256))
    for wgs in sorted({tiles, max(tiles // 2, 1), 1, min(7, tiles * K // 64), min(256, tiles * K // 64), 100}):
        out = torch.full((M, N), float("nan"), device="cuda")
        ops.qgemm_bf16s_nt_sk([(dy, wT, ks, 0.25)], out, wgs=wgs)
        err = float(((out.double() - ref).abs() / den).max())
        assert err < 2e-7, (wgs, err)
        if tiles % wgs == 0:
            assert torch.equal(out, classic), wgs
        for _ in range(3):
            again = torch.full((M, N), float("nan"), device="cuda")
            ops.qgemm_bf16s_nt_sk([(dy, wT, ks, 0.25)], again, wgs=wgs)
            assert torch.equal(out, again), wgs
        base = classic.clone()
        ops.qgemm_bf16s_nt_sk([(dy, wT, ks, 0.25)], base, accumulate=True, wgs=wgs)
        assert rel_err(base.cpu(), 2 * ref.float().cpu()) < 1e-5
    assert ops.nt_sk_error(dy.device) == 0


@pytest.mark.parametrize("mnkk", [(792, 384, 384, 2304), (600, 384, 2304, 384), (515, 192, 192, 1152), (300, 400, 64, 64)])
def test_two_segment_dx_gemm_is_the_sum_of_the_two_gemms(ops, mnkk):
    """[dY_v | dY_qkx] . [W_v ; W_qk]: the input gradients v and W_qk send to x_hat (attention.py:180, :200) as one launch."""
    M, N, K0, K1 = mnkk
    dy0, ks0, w0 = _nt_operands(M, N, K0, 17)
    dy1, ks1, w1 = _nt_operands(M, N, K1, 23)
    t0, t1 = ops.codes_transpose_bf16(w0), ops.codes_transpose_bf16(w1)
    ref = 0.25 * ((dy0.double() * ks0.double()) @ w0.double()) + 0.5 * ((dy1.double() * ks1.double()) @ w1.double())
    den = 0.25 * ((dy0.double() * ks0.double()).abs() @ w0.double().abs()) + 0.5 * ((dy1.double() * ks1.double()).abs() @ w1.double().abs()) + 1e-30
    tiles = ((M + 127) // 128) * ((N + (383 if N > 256 else 255)) // (384 if N > 256 else 256))
    first = None
    for wgs in (tiles, 256, 5):
        out = torch.full((M, N), float("nan"), device="cuda")
        ops.qgemm_bf16s_nt_sk([(dy0, t0, ks0, 0.25), (dy1, t1, None if K1 == 64 else ks1, 0.5)], out, wgs=min(wgs, tiles * (K0 + K1) // 64))
        if K1 == 64:      # second segment without a k-scale vector
            ref_ = 0.25 * ((dy0.double() * ks0.double()) @ w0.double()) + 0.5 * (dy1.double() @ w1.double())
            den_ = 0.25 * ((dy0.double() * ks0.double()).abs() @ w0.double().abs()) + 0.5 * (dy1.double().abs() @ w1.double().abs()) + 1e-30
        else:
            ref_, den_ = ref, den
        assert float(((out.double() - ref_).abs() / den_).max()) < 5e-7, wgs      # fp32 accumulation over 2688 products
        first = out if first is None else first
        assert rel_err(out.cpu(), first.cpu()) < 3e-6          # cut and whole tiles: two associations of the same sums
    assert ops.nt_sk_error(dy0.device) == 0


def test_stream_k_handoff_timeout_is_raised_sticky_and_recoverable(ops):
    """The hand-off of a cut tile (csrc/qgemm_nt_sk.hip, qgemm_bf16s_nt_wide_sk_kernel): a publisher that never sets its flag (fault
    injection word of the workspace) makes the owner's bounded wait run out.  Then: the error word is raised and STAYS raised over
    later launches; ops.nt_sk_poison turns a step's loss into NaN without a host sync; ops.nt_sk_poll raises on the host at the next
    step boundary and re-zeroes the flag area; and the launches after that give the right bits again."""
    M, N, K = 640, 384, 256
    dy, ks, wcodes = _nt_operands(M, N, K, 5)
    wT = ops.codes_transpose_bf16(wcodes)
    dev = dy.device
    good = torch.full((M, N), float("nan"), device="cuda")
    ops.qgemm_bf16s_nt_sk([(dy, wT, ks, 0.25)], good, wgs=3)            # 5 tiles x 4 k-step pairs on 3 workgroups: tiles 1 and 3 are cut
    torch.cuda.synchronize()
    assert ops.nt_sk_error(dev) == 0
    loss = torch.ones((), device="cuda")
    ops.nt_sk_poison(loss)
    assert float(loss) == 1.0
    ops.nt_sk_poll(dev); torch.cuda.synchronize(); ops.nt_sk_poll(dev)       # nothing to report
    ops.nt_sk_inject_fault(dev, 1)                                           # workgroup 1 publishes the head of its run ... never
    bad = torch.full((M, N), float("nan"), device="cuda")
    ops.qgemm_bf16s_nt_sk([(dy, wT, ks, 0.25)], bad, wgs=3)
    torch.cuda.synchronize()
    assert ops.nt_sk_error(dev) != 0
    assert not torch.equal(bad, good)                                       # the owner stored its tile without the partial
    ops.nt_sk_inject_fault(dev, -1)
    again = torch.full((M, N), float("nan"), device="cuda")
    ops.qgemm_bf16s_nt_sk([(dy, wT, ks, 0.25)], again, wgs=3)               # the word is sticky: a clean launch does not clear it
    torch.cuda.synchronize()
    assert ops.nt_sk_error(dev) != 0
    ops.nt_sk_poison(loss)
    assert torch.isnan(loss)
    ops.nt_sk_poll(dev)                                                     # queues the copy ...
    torch.cuda.synchronize()
    with pytest.raises(RuntimeError, match="stream-K hand-off timed out"):
        ops.nt_sk_poll(dev)                                                 # ... and the next step boundary sees it
    torch.cuda.synchronize()
    assert ops.nt_sk_error(dev) == 0                                        # flag area re-zeroed by the poll that raised
    for _ in range(3):
        out = torch.full((M, N), float("nan"), device="cuda")
        ops.qgemm_bf16s_nt_sk([(dy, wT, ks, 0.25)], out, wgs=3)
        assert torch.equal(out, good)
    assert ops.nt_sk_error(dev) == 0


def test_engine_step_surfaces_a_stream_k_timeout(ops):
    """engine.train_step: the step whose backward follows a timed-out hand-off returns a NaN loss, and the next call raises."""
    from ofq_amd import engine
    model = engine.build_student("deit_tiny_distilled_patch16_224", wbits=4, abits=4, depth=1, num_classes=10).cuda()
    img = torch.randn(2, 3, 224, 224, device="cuda")
    tgt = torch.randint(0, 10, (2,), device="cuda")
    soft = torch.randn(2, 10, device="cuda")
    engine.setup_alpha(model, img)
    model.train()
    opt = engine.make_optimizer(model)
    l0 = engine.train_step(model, opt, img, tgt, soft)
    assert torch.isfinite(l0)
    dev = img.device
    # an unrelated stream-K launch on the same stream times out (fault injection) between two steps
    dy, ks, wcodes = _nt_operands(640, 384, 256, 5)
    out = torch.empty((640, 384), device="cuda")
    ops.nt_sk_inject_fault(dev, 1)
    ops.qgemm_bf16s_nt_sk([(dy, ops.codes_transpose_bf16(wcodes), ks, 0.25)], out, wgs=3)
    ops.nt_sk_inject_fault(dev, -1)
    before = [p.detach().clone() for p in model.parameters()]
    l1 = engine.train_step(model, opt, img, tgt, soft)
    torch.cuda.synchronize()
    assert torch.isnan(l1)
    # round 6: the step guard (ofq_step_guard) sits between the backward pass and the optimiser -- nothing was updated
    assert all(torch.equal(a, p.detach()) for a, p in zip(before, model.parameters()))
    with pytest.raises(RuntimeError, match="stream-K hand-off timed out"):
        engine.train_step(model, opt, img, tgt, soft)
        torch.cuda.synchronize()
        engine.train_step(model, opt, img, tgt, soft)
    torch.cuda.synchronize()
    assert ops.nt_sk_error(dev) == 0
    assert torch.isfinite(engine.train_step(model, opt, img, tgt, soft))


@pytest.mark.parametrize("geo", [(3, 3, 224, 224, 16), (2, 3, 224, 224, 4), (2, 3, 64, 96, 8)])
def test_image_quantiser_in_patch_layout_equals_the_permute_copies(ops, geo):
    """ofq_lsq_fwd_patch / _bwd_patch (round 6: the W8A8 stem, qlinear.py:166-174): the image quantiser's values and codes written
    directly in the im2col order of the stride == kernel convolution, its backward reading the GEMM's input gradient in that order.
    Against the plain kernels + the unfold / permute copies the reference makes: every output the same bits."""
    B, Cin, Hh, Ww, p = geo
    g = torch.Generator(device="cuda").manual_seed(B + p)
    x = torch.randn(B, Cin, Hh, Ww, device="cuda", generator=g)
    s = torch.rand(Cin, device="cuda", generator=g) * 0.02 + 0.01
    b4 = torch.randn(Hh * Ww, device="cuda", generator=g) * 0.01
    baft = torch.randn(Hh * Ww, device="cuda", generator=g) * 0.01
    gh, gw, K = Hh // p, Ww // p, Cin * p * p
    plain = ops.LsqGeom(B, Cin, Hh * Ww, Hh * Ww, 0, -128, 127, B * Hh * Ww)
    patch = ops.LsqGeom(B, Cin, Hh * Ww, Hh * Ww, 0, -128, 127, B * Hh * Ww)
    patch.patch = (Ww, p, p)
    im2col = lambda t: t.view(B, Cin, gh, p, gw, p).permute(0, 2, 4, 1, 3, 5).reshape(B * gh * gw, K)      # noqa: E731
    y0, c0 = ops.lsq_fwd(x.view(B * Cin, -1), s, b4, baft, plain, want_codes=True)
    y1, c1 = ops.lsq_fwd(x.view(B * Cin, -1), s, b4, baft, patch, want_codes=True)
    assert torch.equal(y1.view(B * gh * gw, K), im2col(y0)) and torch.equal(c1.view(B * gh * gw, K), im2col(c0))
    assert len(torch.unique(c1)) > 100
    _, c2 = ops.lsq_fwd(x.view(B * Cin, -1), s, b4, baft, patch, want_codes=True, need_values=False)
    assert torch.equal(c2, c1)
    gy = torch.randn(B * gh * gw, K, device="cuda", generator=g) * 1e-2                                   # as the conv's GEMM hands it over
    gy_img = gy.view(B, gh, gw, Cin, p, p).permute(0, 3, 1, 4, 2, 5).reshape(B * Cin, Hh * Ww).contiguous()   # the reference's copy back
    ref = ops.lsq_bwd(gy_img, x.view(B * Cin, -1), s, b4, plain)
    got = ops.lsq_bwd(gy, x.view(B * Cin, -1), s, b4, patch)
    for a, b_ in zip(got, ref):
        assert torch.equal(a, b_)
    if ops.GRAD_PLANES == 2:
        assert ops.amax_of(got[0]) is not None
    # argument checks: widths that are no multiple of 4, a patch that does not tile the image
    bad = ops.LsqGeom(B, Cin, Hh * Ww, Hh * Ww, 0, -128, 127, B * Hh * Ww)
    bad.patch = (Ww, p, p + 1)
    with pytest.raises(RuntimeError):
        ops.lsq_fwd(x.view(B * Cin, -1), s, b4, baft, bad, want_codes=True)


@pytest.mark.parametrize("shift", [0, 3])
def test_layernorm_lsq_with_token_permutations_equals_the_same_kernel_on_permuted_rows(ops, shift):
    """ofq_layernorm_lsq_fwd_perm / _bwd_perm (round 6: Swin's shifted-window partition and reverse folded into the LayerNorm passes,
    swin.py:103-131, :160-170; swin_attention_and_mlp.py:312-323).  The permutation is the real one (7 x 7 windows on a 14 x 14 map,
    cyclic shift 0 / 3).  Reference = the UN-permuted kernels on explicitly permuted rows: every per-row result (codes, x + res, mean,
    rstd, dx, the res gradient, the step gradients' row partials) must be the same bits at its new place; the column sums
    (dgamma, dbeta, d offsets) add the same rows in another order: 1e-5."""
    from ofq_amd.swin import WindowGeometry
    B, Hm, Wm, C = 3, 14, 14, 96
    g = torch.Generator(device="cuda").manual_seed(5 + shift)
    x = torch.randn(B, Hm, Wm, C, device="cuda", generator=g)
    geo = WindowGeometry(x, [7, 7], [shift, shift])
    idx, inv = geo._perm(x.device)
    N, S = Hm * Wm, 49
    x2 = x.reshape(B * N, C)
    res_w = torch.randn(B * N, C, device="cuda", generator=g)                       # window-major
    gam, bet = torch.rand(C, device="cuda", generator=g) + 0.5, torch.randn(C, device="cuda", generator=g) * 0.1
    s = torch.rand(S, device="cuda", generator=g) * 0.2 + 0.3
    b4 = torch.randn(C, device="cuda", generator=g) * 0.05
    geom = ops.LsqGeom(B * N // S, S, C, C, 0, -4, 3, B * N // S * C)
    to_w = lambda t: t.view(B, N, -1)[:, idx.long()].reshape(B * N, -1)              # token-major rows -> window-major   # noqa: E731
    to_t = lambda t: t.view(B, N, -1)[:, inv.long()].reshape(B * N, -1)              # window-major rows -> token-major   # noqa: E731
    # forward
    cw, xsw, mw, rw = ops.layernorm_lsq_fwd(to_w(x2).contiguous(), gam, bet, 1e-5, s, b4, geom, res2d=res_w)
    cp, xsp, mp, rp = ops.layernorm_lsq_fwd(x2, gam, bet, 1e-5, s, b4, geom, res2d=res_w, q_perm=inv, res_perm=inv)
    assert torch.equal(cp, cw)                                                      # codes: window-major in both
    assert torch.equal(to_w(xsp), xsw) and torch.equal(to_w(mp.view(-1, 1)), mw.view(-1, 1)) and torch.equal(to_w(rp.view(-1, 1)), rw.view(-1, 1))
    assert len(torch.unique(cp)) >= 6
    # only the quantised side permuted (norm1: the pending residual is token-major)
    res_t = to_t(res_w).contiguous()
    cq, xsq, _, _ = ops.layernorm_lsq_fwd(x2, gam, bet, 1e-5, s, b4, geom, res2d=res_t, q_perm=inv)
    assert torch.equal(cq, cw) and torch.equal(xsq, xsp)
    # backward
    gq_w = torch.randn(B * N, C, device="cuda", generator=g) * 1e-2                  # gradient of the quantised values: window-major
    dres_t = torch.randn(B * N, C, device="cuda", generator=g) * 1e-2                # gradient arriving on x + res: token-major
    ref = ops.layernorm_lsq_bwd(gq_w, xsw, mw, rw, gam, bet, s, b4, geom, dres2d=to_w(dres_t).contiguous())
    got = ops.layernorm_lsq_bwd(gq_w, xsp, mp, rp, gam, bet, s, b4, geom, dres2d=dres_t, q_perm=inv, res_perm=inv)
    assert len(got) == 7
    assert torch.equal(to_w(got[0]), ref[0])                                        # dx: token-major here, window-major there
    assert torch.equal(got[6], ref[0])                                              # the res gradient: window-major
    assert torch.equal(got[4], ref[4])                                              # ds: the same row partials at the same places
    for i in (1, 2, 3, 5):
        assert rel_err(got[i].cpu(), ref[i].cpu()) < 1e-5, i
    if ops.GRAD_PLANES == 2:
        assert ops.amax_of(got[6]) is not None and ops.amax_of(got[6]) is ops.amax_of(got[0])
    got2 = ops.layernorm_lsq_bwd(gq_w, xsp, mp, rp, gam, bet, s, b4, geom, dres2d=dres_t, q_perm=inv)
    assert got2[6] is None and torch.equal(got2[0], got[0])
    # argument checks: the image length must divide the rows and be a multiple of the step vector
    assert ops.lib().ofq_layernorm_lsq_fwd_perm(x2.data_ptr(), None, None, None, None, None, mp.data_ptr(), rp.data_ptr(), cp.data_ptr(),
                                                s.data_ptr(), S, 1.0, None, -4, 3, B * N, C, C, 1e-5, inv.data_ptr(), None, N - 1,
                                                ops._stream()) != 0


def test_step_guard_words_and_the_guarded_adamw_launch(ops):
    """ofq_step_guard: OR of device words compared as bits without a float's sign bit (an int 1, a float 0.25 from an averaged
    bucket flag, -0.0 = clean, NaN = set); outputs loss / guard word / flag.  ofq_adamw_multi_g / _dev_g: a non-zero guard word
    makes the launch leave p, m, v untouched; a zero word gives the bits of the unguarded launch."""
    from ofq_amd import _lib
    from ofq_amd.optim import FusedAdamW
    dev = torch.device("cuda", 0)
    words = torch.zeros(4, dtype=torch.int32, device=dev)
    fw = words.view(torch.float32)
    guard = torch.full((1,), 7, dtype=torch.int32, device=dev)
    flag = torch.full((1,), 7.0, device=dev)

    def run(n=4, with_loss=True):
        loss = torch.ones(1, device=dev)
        arr = (_lib.vp * 4)(*[words.data_ptr() + 4 * i for i in range(4)])
        ops._chk(ops.lib().ofq_step_guard(arr, n, loss.data_ptr() if with_loss else None, guard.data_ptr(), flag.data_ptr(), ops._stream()), "g")
        torch.cuda.synchronize()
        return float(loss), int(guard), float(flag)
    assert run() == (1.0, 0, 0.0)
    fw[1] = -0.0
    assert run() == (1.0, 0, 0.0)                       # the sign bit alone is not a flag
    words[2] = 1
    l, g, f = run()
    assert l != l and g == 1 and f == 1.0
    assert run(n=2) == (1.0, 0, 0.0)                    # only the first n words count
    words[2] = 0
    fw[0] = 0.25
    l, g, f = run(with_loss=False)
    assert l == 1.0 and g == 1 and f == 1.0             # (no loss pointer: the other outputs still written)
    fw[0] = float("nan")
    assert run()[1] == 1
    fw[0] = 0.0
    assert run() == (1.0, 0, 0.0)
    assert ops.lib().ofq_step_guard(None, 33, None, None, None, ops._stream()) != 0

    torch.manual_seed(0)
    ps = [torch.randn(1000, device=dev), torch.randn(37, 5, device=dev)]
    gs = [torch.randn_like(p) for p in ps]

    def adam(guard_val, captured):
        qs = [torch.nn.Parameter(p.clone()) for p in ps]
        opt = FusedAdamW(qs, lr=1e-2, weight_decay=0.1)
        for q, g_ in zip(qs, gs):
            q.grad = g_.clone()
        opt.step()                                      # creates the state (unguarded: no engine step has made the word yet, or it is 0)
        w = ops.step_guard_word(dev)
        w.fill_(guard_val)
        try:
            if captured:
                opt.begin_capture(dev)
                gph = torch.cuda.CUDAGraph()
                st = torch.cuda.Stream()
                st.wait_stream(torch.cuda.current_stream())
                with torch.cuda.graph(gph, stream=st):
                    opt.step()
                opt.advance_for_replay()
                gph.replay()
            else:
                opt.step()
            torch.cuda.synchronize()
        finally:
            w.zero_()
        return [q.detach().clone() for q in qs] + [opt.state[q]["exp_avg"].clone() for q in qs] + [opt.state[q]["exp_avg_sq"].clone() for q in qs]
    ops.step_guard_word(dev).zero_()
    for captured in (False, True):
        one = adam(0, captured)
        base = adam(0, False)
        assert all(torch.equal(a, b) for a, b in zip(one, base))
        held = adam(1, captured)
        qs = [torch.nn.Parameter(p.clone()) for p in ps]
        opt = FusedAdamW(qs, lr=1e-2, weight_decay=0.1)
        for q, g_ in zip(qs, gs):
            q.grad = g_.clone()
        opt.step()
        torch.cuda.synchronize()
        first = [q.detach().clone() for q in qs] + [opt.state[q]["exp_avg"].clone() for q in qs] + [opt.state[q]["exp_avg_sq"].clone() for q in qs]
        assert all(torch.equal(a, b) for a, b in zip(held, first)), captured      # the guarded second step changed nothing
        assert not all(torch.equal(a, b) for a, b in zip(one, first))


@pytest.mark.parametrize("colmode", [0, 1])
def test_i8_recompute_backward_is_deterministic(ops, colmode):
    """ofq_qgemm_i8_lsq_bwd at the DeiT-S token count, five launches on the same operands: outputs AND the per-workgroup
    partials in the workspace must be bit-identical (an experimental epilogue once made the step-gradient partials of the
    column-mode form vary from launch to launch at rounding level -- the fast / exact choice of a group has to be a
    function of the data alone; lsq.py:593-601 under autograd)."""
    M, N, K, Tn = 128 * 197, 384, 384, 197
    g = torch.Generator(device="cuda").manual_seed(3)
    qa = torch.randint(-4, 4, (M, K), dtype=torch.int8, device="cuda", generator=g)
    qw = (2 * torch.randint(-4, 4, (N, K), device="cuda", generator=g) + 1).to(torch.int8)
    s = torch.rand(Tn, device="cuda", generator=g) * 0.05 + 0.02
    cs = torch.rand(N, device="cuda", generator=g) * 0.05
    bias = torch.randn(N, device="cuda", generator=g) * 0.1
    r = torch.randn(N, device="cuda", generator=g) * 0.1
    qS = N if colmode else Tn
    q = {"s": torch.rand(qS, device="cuda", generator=g) * 0.5 + 0.3, "S": qS, "gscale": 0.01, "b4": bias * 0.5, "lo": -4, "hi": 3,
         "gelu": False, "rowmul": 1, "coldiv": N, "colmode": colmode}
    prod = {"xcodes": qa, "wcodes": qw, "bias": bias, "w_scale": cs, "w_mult": 0.25, "r": r, "act_s": s, "act_S": Tn, "act_gscale": 0.01}
    gy = torch.randn(M, N, device="cuda", generator=g)
    nbytes = ops.lib().ofq_qgemm_i8_lsq_bwd_ws_bytes(M, N, colmode) - 256
    runs = []
    for _ in range(5):
        out = ops.qgemm_i8_lsq_bwd(gy, prod, q)
        torch.cuda.synchronize()
        runs.append(([x.clone() for x in out], ops.workspace(16, gy.device)[:nbytes].clone()))
    for out, ws in runs[1:]:
        assert all(torch.equal(a, b) for a, b in zip(runs[0][0], out))
        assert torch.equal(runs[0][1], ws)


@pytest.mark.parametrize("dims", [(396, 64, 64), (198, 32, 208), (396, 200, 80)])
def test_narrow_tn_kernel_reads_stay_inside_its_operands(ops, dims):
    """Operands narrower than a 128-column tile (the golden attention shapes: 64 channels, 32-channel heads): the staging
    chunks past the last column / row have to be read from inside the matrix.  Both operands sit at the very END of
    their own 32 MiB allocations (the caching allocator gives a request of that size a segment of exactly that size), so
    a read past the last row can leave the mapping -- the way the full GPU suite once caught a dropped clamp, as an
    order-dependent abort; the values are checked as well (qlinear.py:69 under autograd: dW of a quantised linear layer,
    and the same kernel as dV of the attention core)."""
    Ktok, Mo, Nc = dims
    rs = np.random.RandomState(17)
    big_c = torch.empty(32 << 20, dtype=torch.int8, device="cuda")
    big_y = torch.empty(8 << 20, dtype=torch.float32, device="cuda")
    codes = big_c[-Ktok * Nc:].view(Ktok, Nc)
    dy = big_y[-Ktok * Mo:].view(Ktok, Mo)
    codes.copy_(torch.from_numpy(rs.randint(-2, 2, (Ktok, Nc)).astype(np.int8)))
    dy.copy_(T(det_normalish((Ktok, Mo), 31, 1.0)))
    S = 198
    s = T(det_uniform((S,), 33, 0.1, 1.0)).cuda()
    baft = T(det_uniform((Nc,), 34, -0.05, 0.05)).cuda()
    ae = O.lsq_effective_scale(s.cpu(), 0.01)[torch.arange(Ktok) % S].double()
    db = dy.double().sum(0).float()
    ref = (dy.cpu().double() * ae[:, None]).t() @ codes.cpu().double() + db.cpu().double()[:, None] * baft.cpu().double()[None, :]
    den = ((dy.cpu().double() * ae[:, None]).abs().t() @ codes.cpu().double().abs()) + 1e-30
    for split in (1, 3):
        dW = ops.qgemm_bf16s_tn(dy, codes, s, S, 0.01, db, baft, split=split)
        torch.cuda.synchronize()
        assert float(((dW.cpu().double() - ref).abs() / den).max()) < 1e-6, split
    if Mo <= 64 and Nc > 128:        # the dV form of the same kernel (one head: d channels x Np keys), 64 x 256 tiles
        B, H, N, d, Np = 1, 1, Ktok, Mo, Nc
        sp = T(det_uniform((N,), 35, 0.01, 0.1)).cuda()
        dV = ops.qattn_dv(dy.view(B, N, d), codes.view(B, H, N, Np), sp, 0.01, B, H, N, d, Np)
        torch.cuda.synchronize()
        ap = O.lsq_effective_scale(sp.cpu(), 0.01).double()
        want = codes.cpu().double()[:, :N].t() @ (dy.cpu().double() * ap[:, None])
        assert rel_err(dV.cpu().view(N, d), want.float()) < 1e-5


@pytest.mark.parametrize("shape", [(792, 384, 384), (1188, 1536, 384), (396, 384, 1536), (500, 72, 48), (2000, 2304, 384),
                                   (700, 96, 192), (640, 384, 768), (900, 200, 144)])
def test_qgemm_bf16_split_weight_grad_tn(ops, shape):
    Ktok, Mo, Nc = shape
    rs = np.random.RandomState(5)
    dy = T(det_normalish((Ktok, Mo), 91, 1.0)) * T(det_uniform((Ktok, 1), 92, 1e-3, 10.0))
    codes = torch.from_numpy(rs.randint(-8, 8, (Ktok, Nc)).astype(np.int8))
    S = 198 if Ktok % 198 == 0 else Ktok
    s = T(det_uniform((S,), 93, 0.1, 1.0))
    gscale = 0.01
    baft = T(det_uniform((Nc,), 94, -0.05, 0.05))
    db = dy.double().sum(0).float()
    ae = O.lsq_effective_scale(s, gscale)[torch.arange(Ktok) % S].double()
    ref = (dy.double() * ae[:, None]).t() @ codes.double() + db.double()[:, None] * baft.double()[None, :]
    den = ((dy.double() * ae[:, None]).abs().t() @ codes.double().abs()) + 1e-30
    for split in (None, 1, 5):
        dW = ops.qgemm_bf16s_tn(dy.cuda(), codes.cuda(), s.cuda(), S, gscale, db.cuda(), baft.cuda(), split=split)
        err = float(((dW.cpu().double() - ref).abs() / den).max())
        assert err < 1e-6, (split, err)
        dWb, dbg = ops.qgemm_bf16s_tn(dy.cuda(), codes.cuda(), s.cuda(), S, gscale, None, baft.cuda(), split=split,
                                      compute_db=True)
        assert rel_err(dbg.cpu(), db) < 1e-5
        assert float(((dWb.cpu().double() - ref).abs() / den).max()) < 1e-5
    # identical to the fp32-MFMA GEMM on the fake-quant values
    xh = (ae[:, None] * codes.double() + baft.double()).float()
    dW2 = ops.linear_bwd_weight(dy.cuda(), xh.cuda())
    assert rel_err(dW.cpu(), dW2.cpu()) < 1e-5


@pytest.mark.parametrize("cls", [3, 2, 128, 192])
def test_qgemm_tn_group_equals_single_launches(ops, cls):
    """ofq_qgemm_bf16s_tn_group (the deferred weight gradients of a block in one launch) == one ofq_qgemm_bf16s_tn call
    per job with the same split, bit for bit (same partials, same reduction order), dW and db; different token counts,
    step-vector lengths and leading dimensions per job, with and without the offset term."""
    shapes = ([(792, 384, 384), (792, 1536, 384), (594, 384, 1536), (1188, 2304, 384), (396, 128, 768)] if cls == 3 else
              [(792, 576, 192 + 64), (640, 192, 768 + 256), (500, 72, 272)])
    if cls == 192:      # round 6: N in (128, 256) -- DeiT-T's q / k / v / proj / fc1 (N = 192) join grouped launches; the reduce then
        shapes = [(792, 192, 192), (792, 768, 192), (500, 192, 144), (1188, 576, 192), (396, 192, 192 + 48)]     # walks three rows a block
    if cls == 128:      # one DeiT-S block of the headline step: 128 x 197 tokens, the five layers the engine queues together
        shapes = [(25216, 384, 1536), (25216, 1536, 384), (25216, 384, 384), (25216, 384, 384), (25216, 2304, 384)]
    rs = np.random.RandomState(6)
    jobs, refs = [], []
    for i, (Ktok, Mo, Nc) in enumerate(shapes):
        lda = Mo + (8 if i % 2 else 0)
        dyb = (T(det_normalish((Ktok, lda), 191 + i, 1.0)) * T(det_uniform((Ktok, 1), 192 + i, 1e-3, 10.0))).cuda()
        dy = dyb[:, :Mo]
        codes = torch.from_numpy(rs.randint(-8, 8, (Ktok, Nc)).astype(np.int8)).cuda()
        S = 198 if Ktok % 198 == 0 else (197 if Ktok % 197 == 0 else Ktok)
        s = T(det_uniform((S,), 193 + i, 0.1, 1.0)).cuda()
        baft = T(det_uniform((Nc,), 194 + i, -0.05, 0.05)).cuda() if i != 1 else None
        for split in (3,):
            rW, rb = ops.qgemm_bf16s_tn(dy, codes, s, S, 0.01, None, baft, split=split, compute_db=True)
        refs.append((rW, rb))
        jobs.append({"dy2d": dy, "xcodes2d": codes, "lsq_s": s, "S": S, "gscale": 0.01, "baft": baft,
                     "dW": torch.full((Mo, Nc), float("nan"), device="cuda"), "db": torch.full((Mo,), float("nan"), device="cuda")})
    ops.qgemm_bf16s_tn_group(jobs, split=3)
    for j, (rW, rb) in zip(jobs, refs):
        assert torch.equal(j["dW"], rW) and torch.equal(j["db"], rb)
    # default split (about 256 / tiles): against fp64
    for j in jobs:
        j["dW"].fill_(float("nan"))
    ops.qgemm_bf16s_tn_group(jobs)
    for j, (rW, rb) in zip(jobs, refs):
        assert rel_err(j["dW"].cpu(), rW.cpu()) < 1e-5 and rel_err(j["db"].cpu(), rb.cpu()) < 1e-5


def test_qattn_dqkx_persistent_stream_any_tiles_per_workgroup(ops, monkeypatch):
    """ofq_qattn_dqkx_bf16s at DeiT-S dimensions runs on persistent workgroups that walk several (head, m-tile) tiles of
    an image as one k-step stream (qgemm_bf16s_tn_wide_stream_kernel).  The result must not depend on how many tiles a
    workgroup walks (1, 2, 4, 12 of the 12 tiles of an image; 5 is rounded down to a divisor): bit-identical outputs, and
    the fp32-exact product of (dS * a_eff) with the codes plus the offset term (attention.py:210 autograd) against fp64."""
    B, H, N, C = 3, 6, 198, 384
    Np = 208
    rs = np.random.RandomState(11)
    dS = torch.from_numpy(rs.randn(B, H, N, Np).astype(np.float32)) * T(det_uniform((B, H, N, 1), 301, 1e-3, 10.0))
    xcodes = torch.from_numpy(rs.randint(-2, 2, (B, N, C)).astype(np.int8))
    sx = T(det_uniform((N,), 302, 0.1, 1.0))
    bax = T(det_uniform((C,), 303, -0.05, 0.05))
    gx = 0.01
    ae = O.lsq_effective_scale(sx, gx).double()
    d = dS[..., :N].double()                                              # [b,h,n,m]
    ref = torch.einsum("bhnm,bnc->bmhc", d * ae[None, None, :, None], xcodes.double()) \
        + d.sum(2).permute(0, 2, 1)[..., None] * bax.double()[None, None, None, :]
    den = torch.einsum("bhnm,bnc->bmhc", (d * ae[None, None, :, None]).abs(), xcodes.double().abs()) + 1e-30
    outs = []
    for tpw in ("1", "2", "4", "12", "5"):
        monkeypatch.setenv("OFQ_TN_STREAM_TPW", tpw)
        dq = ops.qattn_dqkx(dS.cuda(), xcodes.cuda(), sx.cuda(), gx, bax.cuda(), B, H, N, C, Np)
        outs.append(dq.cpu())
        assert float(((dq.cpu().double() - ref).abs() / den).max()) < 2e-6, tpw
    monkeypatch.delenv("OFQ_TN_STREAM_TPW")
    dq = ops.qattn_dqkx(dS.cuda(), xcodes.cuda(), sx.cuda(), gx, bax.cuda(), B, H, N, C, Np)
    for o in outs:
        assert torch.equal(o, dq.cpu())


@pytest.mark.parametrize("dims", [(3, 6, 198, 64, 208, 3), (2, 3, 198, 64, 208, 15), (2, 2, 250, 32, 256, 3), (2, 4, 100, 48, 112, 7),
                                  (128, 6, 197, 64, 208, 3)])
def test_qattn_dp_softmax_bwd_fused_equals_the_two_kernels(ops, dims):
    """ofq_qattn_dp_softmax_bwd (dP GEMM + softmax-LSQ backward, dP never in memory) against ofq_qattn_dp followed by
    ofq_softmax_lsq_bwd: dS, the step gradient ds and the row sums; and against fp64 of the same formulas
    (attention.py:213-219 under autograd).  Head dims 64 / 32 / 48, 2- and 4-bit codes, N up to 250 keys."""
    B, H, N, d, Np, hi = dims
    C = H * d
    rs = np.random.RandomState(21)
    dO = torch.from_numpy(rs.randn(B, N, C).astype(np.float32)) * T(det_uniform((B, N, 1), 401, 1e-3, 3.0))
    vcodes = torch.from_numpy(rs.randint(-4, 4, (B, N, C)).astype(np.int8))
    sv = T(det_uniform((C,), 402, 0.05, 0.5))
    bav = T(det_uniform((C,), 403, -0.05, 0.05))
    gv = 0.02
    logits = torch.from_numpy(rs.randn(B, H, N, N).astype(np.float32)) * 2.0
    prob = torch.zeros(B, H, N, Np)
    prob[..., :N] = torch.softmax(logits, -1)
    sm_s = T(det_uniform((N,), 404, 0.02, 0.2))
    alpha = d ** -0.5
    rows = B * H * N
    cu = lambda t: t.cuda()
    w = ops.rowdot_f32_seg(cu(dO).view(B * N, C), cu(bav), H, d)
    dP = ops.qattn_dp(cu(dO), cu(vcodes), cu(sv), gv, w, B, H, N, d, Np)
    dS0, ds0, rs0 = ops.softmax_lsq_bwd(dP.clone(), cu(prob), cu(sm_s), rows, N, Np, N, alpha, hi, rows, inplace=True,
                                        want_rowsum=True)
    dS1, ds1, rs1 = ops.qattn_dp_softmax_bwd(cu(dO), cu(vcodes), cu(sv), gv, cu(bav), cu(prob), cu(sm_s), alpha, hi, B, H, N, d,
                                             Np, want_rowsum=True)
    assert rel_err(dS1[..., :N].cpu(), dS0[..., :N].cpu()) < 2e-6
    assert rel_err(ds1.cpu(), ds0.cpu()) < 1e-5
    assert float((rs1.cpu() - rs0.cpu()).abs().max()) < 1e-5 * float(dS0[..., :N].abs().max())
    assert float(dS1[..., N:].abs().max()) == 0.0 if Np > N else True
    # fp64 of the formulas
    ae = O.lsq_effective_scale(sv, gv).double()
    vh = vcodes.double() * ae + bav.double()
    dPr = torch.einsum("bnhj,bmhj->bhnm", dO.double().view(B, N, H, d), vh.view(B, N, H, d))
    a = O.lsq_effective_scale(sm_s, 1.0 / (hi * rows) ** 0.5).double()[None, None, :, None]
    p = prob[..., :N].double()
    v = p / a
    inr = (v >= 0) & (v <= hi)
    dq = torch.where(inr, dPr, torch.zeros_like(dPr))
    dot = (dq * p).sum(-1, keepdim=True)
    ref = (dq - dot) * p * alpha
    assert rel_err(dS1[..., :N].cpu(), ref.float()) < 1e-5


# ------------------------------------------------------------------------------------------------ attention on codes
def test_qattn_code_kernels_vs_fp64(ops):
    B, H, N, d = 2, 3, 198, 32
    C = H * d
    Np = 208
    rs = np.random.RandomState(7)
    gx, gq, gp, gv = 0.011, 0.013, 0.017, 0.019
    xcodes = torch.from_numpy(rs.randint(-2, 2, (B, N, C)).astype(np.int8))
    qcodes = torch.from_numpy(rs.randint(-2, 2, (B, N, H, C)).astype(np.int8))
    vcodes = torch.from_numpy(rs.randint(-8, 8, (B, N, C)).astype(np.int8))
    pcodes = torch.zeros(B, H, N, Np, dtype=torch.uint8)
    pcodes[..., :N] = torch.from_numpy(rs.randint(0, 4, (B, H, N, N)).astype(np.uint8))
    sx, sq = T(det_uniform((N,), 101, 0.2, 1.0)), T(det_uniform((N * H,), 102, 0.2, 1.0))
    sp, sv = T(det_uniform((N,), 103, 0.05, 0.3)), T(det_uniform((C,), 104, 0.2, 1.0))
    bax, baq, bav = T(det_uniform((C,), 105, -0.1, 0.1)), T(det_uniform((H * C,), 106, -0.1, 0.1)), T(det_uniform((C,), 107, -0.1, 0.1))
    ax, aq = O.lsq_effective_scale(sx, gx).double(), O.lsq_effective_scale(sq, gq).double().view(N, H)
    ap, av = O.lsq_effective_scale(sp, gp).double(), O.lsq_effective_scale(sv, gv).double()
    xh = ax[None, :, None] * xcodes.double() + bax.double()                                   # (B,N,C)
    qh = aq[None, :, :, None] * qcodes.double() + baq.double().view(1, 1, H, C)               # (B,N,H,C)
    vh = av * vcodes.double() + bav.double()                                                  # (B,N,C)
    ph = ap[None, None, :, None] * pcodes[..., :N].double()                                   # (B,H,N,N)
    cu = lambda t: t.cuda()
    # ---- scores
    u = ops.rowdot_i8_multi(cu(xcodes).view(B * N, C), cu(baq).view(H, C))
    assert rel_err(u.cpu(), (xcodes.double().view(B * N, C) @ baq.double().view(H, C).t()).float()) < 1e-6
    tq = ops.rowdot_i8(cu(qcodes).view(B * N * H, C), cu(bax))
    z = (baq.double().view(H, C) @ bax.double()).float()
    S = ops.qattn_scores(cu(xcodes), cu(qcodes), cu(sx), gx, cu(sq), gq, u, tq, cu(z), B, H, N, C, Np)
    S_ref = torch.einsum("bnc,bmhc->bhnm", xh, qh)
    assert rel_err(S.cpu()[..., :N], S_ref.float()) < 2e-6
    # ---- P.V
    vT = ops.codes_transpose_i8(cu(vcodes), Np)
    assert torch.equal(vT.cpu()[:, :, :N], vcodes.transpose(1, 2)) and int(vT.cpu()[:, :, N:].abs().max()) == 0
    rp = pcodes.float().sum(-1).reshape(-1)
    Oo = ops.qattn_pv(cu(pcodes).view(torch.int8), vT, cu(sp), gp, cu(sv), gv, cu(bav), cu(rp), B, H, N, d, Np)
    O_ref = (ph @ vh.view(B, N, H, d).permute(0, 2, 1, 3)).transpose(1, 2).reshape(B, N, C)
    assert rel_err(Oo.cpu(), O_ref.float()) < 2e-6
    # ---- dP, dV
    dO = T(det_normalish((B, N, C), 108, 1.0))
    w = ops.rowdot_f32_seg(cu(dO).view(B * N, C), cu(bav), H, d)
    dP = ops.qattn_dp(cu(dO), cu(vcodes), cu(av.float()), 0.0, w, B, H, N, d, Np)          # effective step given
    assert torch.equal(dP[..., :N], ops.qattn_dp(cu(dO), cu(vcodes), cu(sv), gv, w, B, H, N, d, Np)[..., :N])   # ... or in-kernel
    dP_ref = torch.einsum("bnhj,bmhj->bhnm", dO.double().view(B, N, H, d), vh.view(B, N, H, d))
    assert rel_err(dP.cpu()[..., :N], dP_ref.float()) < 2e-6
    dV = ops.qattn_dv(cu(dO), cu(pcodes).view(torch.int8), cu(sp), gp, B, H, N, d, Np)
    dV_ref = torch.einsum("bhnm,bnhj->bmhj", ph, dO.double().view(B, N, H, d)).reshape(B, N, C)
    assert rel_err(dV.cpu(), dV_ref.float()) < 2e-6
    # ---- dqkx, dxq
    dS = torch.zeros(B, H, N, Np)
    dS[..., :N] = T(det_normalish((B, H, N, N), 109, 1.0))
    dq = ops.qattn_dqkx(cu(dS), cu(xcodes), cu(sx), gx, cu(bax), B, H, N, C, Np)
    dq_ref = torch.einsum("bhnm,bnc->bmhc", dS[..., :N].double(), xh)
    assert rel_err(dq.cpu(), dq_ref.float()) < 2e-6
    dxq = ops.qattn_dxq(cu(dS), cu(qcodes), cu(sq), gq, B, H, N, C, Np)
    dxq_ref = torch.einsum("bhnm,bmhc->bnc", dS[..., :N].double(), aq[None, :, :, None] * qcodes.double())
    assert rel_err(dxq.cpu(), dxq_ref.float()) < 2e-6
    dxq2 = ops.qattn_dxq(cu(dS), cu(qcodes), cu(sq), gq, B, H, N, C, Np, out=dxq.clone(), accumulate=True)
    assert rel_err(dxq2.cpu(), 2 * dxq_ref.float()) < 2e-6


def test_softmax_lsq_codes_and_rowsums(ops):
    B, H, N, bits = 2, 3, 198, 2
    ld = 208
    hi = 3
    sc = torch.zeros(B, H, N, ld)
    sc[..., :N] = T(det_normalish((B, H, N, N), 111, 3.0))
    s = T(det_uniform((N,), 112, 0.01, 0.05))
    M = B * H * N
    prob, y, codes, rsum = ops.softmax_lsq_fwd(sc.cuda(), s.cuda(), B * H * N, N, ld, N, 0.125, hi, M, want_codes=True)
    a = O.lsq_effective_scale(s, 1.0 / math.sqrt(hi * M))
    q_ref = torch.clamp(prob.cpu()[..., :N] / a[:, None], 0, hi).round()
    assert torch.equal(codes.cpu()[..., :N].float(), q_ref)
    assert int(codes.cpu()[..., N:].max()) == 0
    assert torch.equal(rsum.cpu().view(B, H, N), q_ref.sum(-1))
    g = torch.zeros(B, H, N, ld)
    g[..., :N] = T(det_uniform((B, H, N, N), 113, -1, 1))
    dS, ds, rs = ops.softmax_lsq_bwd(g.cuda(), prob, s.cuda(), B * H * N, N, ld, N, 0.125, hi, M, inplace=False, want_rowsum=True)
    assert torch.allclose(rs.cpu().view(B, H, N), dS.cpu().sum(-1), atol=1e-6)
    assert float(dS.cpu()[..., N:].abs().max()) == 0.0


# ------------------------------------------------------------------------------------------------ LayerNorm
@pytest.mark.parametrize("shape", [(2 * 198, 384), (3 * 197, 192), (50, 768), (33, 96), (7, 1536), (5, 2048), (300, 100)])
@pytest.mark.parametrize("fused_add", [False, True])
def test_layernorm_fwd_bwd_vs_fp64(ops, shape, fused_add):
    """csrc/layernorm.hip against nn.LayerNorm semantics (deit_vision_transformer.py Block.norm1/norm2) in fp64;
    tolerance 2e-6 relative on the outputs, 1e-5 on gradients (fp32 row reductions in a different order)."""
    R, C = shape
    x = T(det_normalish((R, C), 301, 1.0)) * 3.0 + 0.5
    res = T(det_normalish((R, C), 302, 1.0)) if fused_add else None
    gamma = T(det_uniform((C,), 303, 0.5, 1.5))
    beta = T(det_uniform((C,), 304, -0.2, 0.2))
    dy = T(det_normalish((R, C), 305, 1.0)) * 1e-2
    dres = T(det_normalish((R, C), 306, 1.0)) * 1e-2 if fused_add else None
    eps = 1e-6
    y, xs, mean, rstd = ops.layernorm_fwd(x.cuda(), gamma.cuda(), beta.cuda(), eps, res2d=None if res is None else res.cuda())
    xd = (x.double() + (res.double() if fused_add else 0.0)).requires_grad_(True)
    gd, bd = gamma.double().requires_grad_(True), beta.double().requires_grad_(True)
    ref = torch.nn.functional.layer_norm(xd, (C,), gd, bd, eps)
    assert rel_err(y.cpu(), ref.detach().float()) < 2e-6
    if fused_add:
        assert torch.equal(xs.cpu(), x + res)
    ref.backward(dy.double())
    xin = xs if fused_add else x.cuda()
    dx, dg, db = ops.layernorm_bwd(dy.cuda(), xin, mean, rstd, gamma.cuda(), dres2d=None if dres is None else dres.cuda())
    dxr = xd.grad + (dres.double() if fused_add else 0.0)
    assert rel_err(dx.cpu(), dxr.float()) < 1e-5
    assert rel_err(dg.cpu(), gd.grad.float()) < 1e-5
    assert rel_err(db.cpu(), bd.grad.float()) < 1e-5


def test_layernorm_autograd_functions_match_torch(ops):
    from ofq_amd import functional as Fq
    torch.manual_seed(0)
    B, N, C = 4, 197, 384
    norm = torch.nn.LayerNorm(C, eps=1e-6).cuda()
    with torch.no_grad():
        norm.weight.uniform_(0.5, 1.5)
        norm.bias.uniform_(-0.1, 0.1)
    x = torch.randn(B, N, C, device="cuda", requires_grad=True)
    r = torch.randn(B, N, C, device="cuda", requires_grad=True)
    w = torch.randn(B, N, C, device="cuda")
    xs, y = Fq.add_layer_norm(norm, x, r)
    (xs * w + y * w.flip(0)).sum().backward()
    got = [x.grad.clone(), r.grad.clone(), norm.weight.grad.clone(), norm.bias.grad.clone()]
    for t in (x, r, norm.weight, norm.bias):
        t.grad = None
    xs2 = x + r
    y2 = norm(xs2)
    (xs2 * w + y2 * w.flip(0)).sum().backward()
    assert rel_err(y, y2) < 2e-6 and torch.equal(xs, xs2)
    for a, b in zip(got, [x.grad, r.grad, norm.weight.grad, norm.bias.grad]):
        assert rel_err(a, b) < 1e-5
    # plain LayerNorm, non-affine
    norm2 = torch.nn.LayerNorm(C, eps=1e-5, elementwise_affine=False).cuda()
    x.grad = None
    y = Fq.layer_norm(norm2, x)
    (y * w).sum().backward()
    g1 = x.grad.clone()
    x.grad = None
    (norm2(x) * w).sum().backward()
    assert rel_err(g1, x.grad) < 1e-5


# ------------------------------------------------------------------------------------------------ AdamW (+ CGA freeze)
def test_fused_adamw_matches_torch_adamw_and_cga_sequence(ops):
    """ofq_adamw_multi against torch.optim.AdamW on the CPU (the optimiser the reference gets from timm, train.py:662)
    over several steps with the two timm parameter groups; with a CGA mask it must equal the reference's
    mask-grad / step / restore sequence (cga.py:962-964, :986, :994-997).  Tolerance 2e-6 (fp32 op order)."""
    from ofq_amd.optim import FusedAdamW
    torch.manual_seed(0)
    shapes = [(384, 384), (1536, 384), (384,), (7,), (3, 5, 16, 16), (100003,)]
    cpu = [torch.randn(*s) for s in shapes]
    ref_p = [t.clone().requires_grad_(True) for t in cpu]
    dev_p = [t.clone().cuda().requires_grad_(True) for t in cpu]
    groups = lambda ps: [{"params": [p for p in ps if p.dim() <= 1], "weight_decay": 0.0},        # noqa: E731
                         {"params": [p for p in ps if p.dim() > 1], "weight_decay": 0.05}]
    ref = torch.optim.AdamW(groups(ref_p), lr=5.47e-4, weight_decay=0.05)
    opt = FusedAdamW(groups(dev_p), lr=5.47e-4, weight_decay=0.05)
    frz_cpu = (torch.rand(384, 384) < 0.3).float()
    for it in range(4):
        lr = 5.47e-4 * (1.0 - 0.1 * it)
        for g1, g2 in zip(ref.param_groups, opt.param_groups):
            g1["lr"] = g2["lr"] = lr
        grads = [torch.randn(*s) * (0.1 if it % 2 else 1e-3) for s in shapes]
        for p, q, g in zip(ref_p, dev_p, grads):
            p.grad = g.clone()
            q.grad = g.clone().cuda()
        use_cga = it >= 2
        if use_cga:       # reference sequence on the CPU side
            ref_p[0].grad.mul_(1.0 - frz_cpu)
            saved = ref_p[0].detach() * frz_cpu
            opt.set_frozen(dev_p[0], frz_cpu.cuda())
        ref.step()
        opt.step()
        if use_cga:
            with torch.no_grad():
                ref_p[0].copy_(ref_p[0] * (1.0 - frz_cpu) + saved)
            opt.clear_frozen()
        for p, q in zip(ref_p, dev_p):
            assert rel_err(q.detach().cpu(), p.detach()) < 2e-6
    for p, q in zip(ref_p, dev_p):
        assert rel_err(opt.state[q]["exp_avg"].cpu(), ref.state[p]["exp_avg"]) < 2e-6
        assert rel_err(opt.state[q]["exp_avg_sq"].cpu(), ref.state[p]["exp_avg_sq"]) < 2e-6
        assert float(opt.state[q]["step"]) == float(ref.state[p]["step"]) == 4.0
    # frozen weights kept their value bit for bit through the two CGA steps
    # state dict layout is torch.optim.AdamW's: it loads into one and back
    other = torch.optim.AdamW(groups([t.clone().cuda().requires_grad_(True) for t in cpu]), lr=1e-3)
    other.load_state_dict(opt.state_dict())
    opt.load_state_dict(other.state_dict())


def _adversarial_level_pairs(lo, hi, count, seed):
    """(step a, input x) pairs, x within 3 ulp of a tie (k + 0.5) * a, for which rint(clip(fl(x * fl(1/a)))) differs from
    the reference's rint(clip(fl(x / a))) -- the cases a reciprocal-multiply shortcut gets wrong -- padded with plain ties."""
    rng = np.random.RandomState(seed)
    n = 400000
    a = (0.05 + rng.rand(n)).astype(np.float32)
    k = rng.randint(lo - 1, hi + 2, size=n).astype(np.float32) + np.float32(0.5)
    x = (k * a).astype(np.float32)
    u = rng.randint(-3, 4, size=n)
    for d in range(1, 4):
        x[u >= d] = np.nextafter(x[u >= d], np.float32(np.inf))
        x[u <= -d] = np.nextafter(x[u <= -d], np.float32(-np.inf))
    ra = (np.float32(1) / a).astype(np.float32)
    fast = np.rint(np.clip((x * ra).astype(np.float32), lo, hi))
    exact = np.rint(np.clip((x / a).astype(np.float32), lo, hi))
    bad = np.nonzero(fast != exact)[0]
    assert bad.size >= count // 2, "the search should find plenty of disagreeing pairs"
    idx = np.concatenate([bad[:count], np.arange(count)])[:count]
    return a[idx], x[idx], exact[idx]


@pytest.mark.gpu
@pytest.mark.parametrize("colmode", [0, 1])
@pytest.mark.parametrize("bits,signed", [(2, True), (2, False), (4, True), (8, False)])
def test_i8_epilogue_levels_on_rounding_boundaries(ops, bits, signed, colmode):
    """The fused next-layer codes of the int8 GEMM (interior-tile epilogue: reciprocal multiply, exact division only for
    waves that flag a product next to a half-integer) must equal rint(clip(fl(x / a))) exactly where that matters: every
    row (row mode) / column (column mode) of this tile carries a (step, input) pair on which x * (1/a) and x / a round to
    different levels (qlinear.py:66-68, lsq.py:593-601)."""
    M = N = 128
    K = 16
    lo, hi = (-(2 ** (bits - 1)), 2 ** (bits - 1) - 1) if signed else (0, 2 ** bits - 1)
    a, x, want1 = _adversarial_level_pairs(lo, hi, 128, 40 + bits + 10 * colmode)
    xc = torch.zeros((M, K), dtype=torch.int8, device="cuda")
    wc = torch.zeros((N, K), dtype=torch.int8, device="cuda")
    ones = torch.ones(128, device="cuda")
    fuse = dict(s=T(a).cuda(), S=128, gscale=0.0, b4=None, lo=lo, hi=hi, gelu=0, colmode=colmode)
    if colmode:      # y[m][n] = r[n] = x[n] (zero codes, unit scales)
        y = ops.qgemm_i8_nt(xc, wc, None, ones, 1.0, T(x).cuda(), ones, M, 0.0, fuse=fuse)
        want_y = np.broadcast_to(x[None, :], (M, N))
        want = np.broadcast_to(want1[None, :], (M, N))
    else:            # y[m][n] = a_eff[m] * (+-1) = x[m]: the row's input step carries |x|, the code product its sign
        xc[:, 0] = torch.from_numpy(np.where(x < 0, -1, 1).astype(np.int8)).cuda()
        wc[:, 0] = 1
        y = ops.qgemm_i8_nt(xc, wc, None, ones, 1.0, None, T(np.abs(x)).cuda(), M, 0.0, fuse=fuse)
        want_y = np.broadcast_to(x[:, None], (M, N))
        want = np.broadcast_to(want1[:, None], (M, N))
    assert np.array_equal(y.cpu().numpy(), want_y)
    got = fuse["codes_out"].cpu().numpy().astype(np.float32)
    if not signed and bits == 8:
        got = np.where(got < 0, got + 256, got)                       # uint8 levels travel in an int8 tensor
    assert np.array_equal(got, want)


@pytest.mark.gpu
@pytest.mark.parametrize("colmode", [0, 1])
def test_i8_epilogue_gelu_levels_next_to_rounding_boundaries(ops, colmode):
    """fc1's epilogue decides fc2's input level on a cheap GELU (Abramowitz-Stegun erf on v_rcp / v_exp) and redoes a
    group with erff + the IEEE division when the result lies within the approximation's error bound of a half-integer.
    Every (pre-activation, step, offset) triple here puts (gelu(y) + b) / a within 1e-8 .. 1e-4 of a rounding boundary;
    the interior-tile codes must equal those of an edge tile of the same kernel, whose epilogue is the plain exact form
    (qlinear.py:128-134: act -> LSQ of fc2's input)."""
    lo, hi = -2, 1
    rng = np.random.RandomState(77 + colmode)
    K = 16
    for rep in range(6):
        n = 1536 if colmode else 128
        yv = rng.uniform(-7.0, 7.0, n)
        yv[: n // 4] = rng.normal(0, 1.2, n // 4)
        a = rng.uniform(0.03, 0.6, n)
        g = 0.5 * yv * (1.0 + np.vectorize(__import__("math").erf)(yv / np.sqrt(2.0)))
        k = rng.randint(lo, hi, n) + 0.5                                     # boundaries lo+.5 .. hi-.5
        delta = np.sign(rng.randn(n)) * 10.0 ** rng.uniform(-8, -4, n)
        if colmode:          # y[m][n] = r[n]; per-column step a[n] and offset b[n] = a (k + .5) - gelu(y) + delta
            b4 = (a * k - g + delta).astype(np.float32)
            ones = torch.ones(128, device="cuda")
            outs = []
            for M in (128, 120):
                xc = torch.zeros((M, K), dtype=torch.int8, device="cuda")
                wc = torch.zeros((n, K), dtype=torch.int8, device="cuda")
                fuse = dict(s=T(a.astype(np.float32)).cuda(), S=n, gscale=0.0, b4=T(b4).cuda(), lo=lo, hi=hi, gelu=1, colmode=1)
                y = ops.qgemm_i8_nt(xc, wc, None, torch.ones(n, device="cuda"), 1.0, T(yv.astype(np.float32)).cuda(), ones[:M], M,
                                    0.0, fuse=fuse)
                assert np.array_equal(y.cpu().numpy()[0], yv.astype(np.float32))
                outs.append(fuse["codes_out"].cpu().numpy())
            assert np.array_equal(outs[0][:120], outs[1])
            assert len(np.unique(outs[0])) == hi - lo + 1
        else:                # y[m][n] = a_eff[m] * (+-1) = y[m]; per-row step a[m] = (gelu(y) + b) / (k + .5) * (1 + delta)
            b = 2.0
            a = (g + b) / (rng.randint(0, 3, n) + 0.5) * (1.0 + delta)                 # boundaries 0.5, 1.5, 2.5 of levels -2 .. 3
            outs = []
            for N in (128, 112):
                xc = torch.zeros((128, K), dtype=torch.int8, device="cuda")
                wc = torch.zeros((N, K), dtype=torch.int8, device="cuda")
                xc[:, 0] = torch.from_numpy(np.where(yv < 0, -1, 1).astype(np.int8)).cuda()
                wc[:, 0] = 1
                fuse = dict(s=T(a.astype(np.float32)).cuda(), S=128, gscale=0.0, b4=torch.full((N,), b, device="cuda"), lo=lo, hi=hi + 2,
                            gelu=1, colmode=0)
                ops.qgemm_i8_nt(xc, wc, None, torch.ones(N, device="cuda"), 1.0, None, T(np.abs(yv).astype(np.float32)).cuda(), 128, 0.0,
                                fuse=fuse)
                outs.append(fuse["codes_out"].cpu().numpy())
            assert np.array_equal(outs[0][:, :112], outs[1])


@pytest.mark.gpu
def test_statsq_multi_tensor_equals_the_per_tensor_launch(ops):
    """ofq_statsq_codes_multi (all layers' weight operands in one or two launches) against ofq_statsq_codes_fwd, bit for
    bit: scale, odd int8 codes, transposed bf16 codes, offset row-dots; 45 tensors so that the table is split over two
    launches; shapes of the DeiT-S / DeiT-T layers plus ragged ones, 2-4 bits, with and without rvec / transposed codes."""
    rng = np.random.RandomState(5)
    shapes = [(384, 384), (1536, 384), (384, 1536), (192, 192), (576, 192), (10, 20), (7, 33), (130, 64), (96, 288)]
    items = []
    for i in range(45):
        r, c = shapes[i % len(shapes)]
        W = T(rng.randn(r, c).astype(np.float32) * 0.05).cuda()
        bits = 2 + i % 3
        rvec = T(rng.randn(c).astype(np.float32) * 0.1).cuda() if i % 4 else None
        items.append((W, bits, rvec, bool(i % 5)))
    got = ops.statsq_codes_multi(items)
    for (W, bits, rvec, want_T), (s, codes, codesT, r) in zip(items, got):
        _, s1, c1, t1, r1 = ops.statsq_codes_fwd(W, bits, rvec=rvec, need_values=False, want_T=want_T)
        assert torch.equal(s, s1) and torch.equal(codes, c1)
        assert (codesT is None) == (t1 is None) and (codesT is None or torch.equal(codesT, t1))
        assert (r is None) == (r1 is None) and (r is None or torch.equal(r, r1))


@pytest.mark.gpu
@pytest.mark.parametrize("N", [198, 130, 64])
def test_qattn_dxq_wide_kernel_vs_fp64(ops, N):
    """dxq on the 128x384-tile kernel (C = 384: the DeiT-S case; k runs over (head, key token) with a k tail since
    N % 32 != 0, rows past N in the second token tile, pad columns of dS left uninitialised on purpose) against fp64,
    plain and accumulating (attention.py:210, backward of x_hat . qkx_hat^T)."""
    B, H, C = 3, 6, 384
    Np = (N + 7) // 8 * 8
    rs = np.random.RandomState(11 + N)
    gq = 0.013
    qcodes = torch.from_numpy(rs.randint(-2, 2, (B, N, H, C)).astype(np.int8))
    sq = T(0.05 + rs.rand(N * H).astype(np.float32))
    dS = torch.full((B, H, N, Np), float("nan"))
    dS[..., :N] = T(rs.randn(B, H, N, N).astype(np.float32) * 1e-2)
    aq = O.lsq_effective_scale(sq, gq).view(N, H).double()
    ref = torch.einsum("bhnm,bmhc->bnc", dS[..., :N].double(), aq[None, :, :, None] * qcodes.double())
    cu = lambda t: t.cuda()
    dxq = ops.qattn_dxq(cu(dS), cu(qcodes), cu(sq), gq, B, H, N, C, Np)
    assert rel_err(dxq.cpu(), ref.float()) < 2e-6
    base = T(rs.randn(B, N, C).astype(np.float32))
    dxq2 = ops.qattn_dxq(cu(dS), cu(qcodes), cu(sq), gq, B, H, N, C, Np, out=base.clone().cuda(), accumulate=True)
    assert rel_err(dxq2.cpu(), (ref + base.double()).float()) < 2e-6


@pytest.mark.gpu
@pytest.mark.parametrize("case", [
    # (M, N, K, S, mode, rowmul, gelu, lo, hi)
    ("row", 2 * 198, 384, 64, 198, 0, 1, 0, -2, 1),            # proj/fc1-like, per-token step
    ("row_gelu", 3 * 50, 256, 48, 50, 0, 1, 1, 0, 3),          # fc1 -> fc2's unsigned quantiser behind the GELU; ragged M
    ("heads", 2 * 70, 3 * 128, 32, 70, 0, 3, 0, -4, 3),        # qkx-like: 3 heads side by side, step per (token, head)
    ("col", 2 * 198, 192, 64, 198, 1, 1, 0, -2, 1),            # v-like: per-channel step; N not a multiple of 128
    ("row_wide_range", 16 * 256, 512, 32, 256, 0, 1, 0, -8, 7),  # 2M elements, gradients over 9 decades: the division
    ("col_wide_range", 16 * 256, 256, 32, 256, 1, 1, 0, -2, 1),  # shortcut (ofq_lsq_bwd_fast) against the IEEE sequence
    ("qkx_full", 128 * 197, 6 * 384, 384, 197, 0, 6, 0, -2, 1),  # the headline step's qkx tensor: 25 216 tokens x 6 heads
    ("v_full", 128 * 197, 384, 384, 197, 1, 1, 0, -2, 1),        # and its v tensor (per-channel step)
])
def test_i8_recompute_lsq_backward_equals_stored_activation_pair(ops, case):
    """ofq_qgemm_i8_lsq_bwd (layer output recomputed from the integer codes, consumer quantiser's backward in registers)
    against the pair it replaces: ofq_qgemm_i8_nt storing y, then ofq_lsq_bwd on the stored y (qlinear.py:58-73 +
    lsq.py:571-602).  The codes-only forward must emit the same codes; dy bit for bit (same expression on the same
    values); the reduced gradients to 1e-5 (different summation order)."""
    name, M, N, K, S, colmode, rowmul, gelu, lo, hi = case
    g = torch.Generator(device="cuda").manual_seed(len(name) * 7 + M)
    xc = torch.randint(-2, 2, (M, K), dtype=torch.int8, device="cuda", generator=g)
    wc = (2 * torch.randint(-2, 2, (N, K), device="cuda", generator=g) + 1).to(torch.int8)
    bias = torch.randn(N, device="cuda", generator=g) * 0.1
    cs = torch.rand(N, device="cuda", generator=g) * 0.05 + 0.01
    r = torch.randn(N, device="cuda", generator=g) * 0.3
    s_in = torch.rand(S, device="cuda", generator=g) * 0.3 + 0.05
    gscale_in = 0.013
    T_tok = S
    coldiv = N // rowmul
    if colmode:
        qs = torch.rand(N, device="cuda", generator=g) * 0.3 + 0.1
        qS = N
    else:
        qs = torch.rand(T_tok * rowmul, device="cuda", generator=g) * 0.3 + 0.1
        qS = T_tok * rowmul
    qb4 = torch.randn(N, device="cuda", generator=g) * 0.05
    qg = 0.021
    gy = torch.randn(M, N, device="cuda", generator=g)
    if "wide_range" in name:
        gy = gy * torch.pow(10.0, torch.randint(-6, 4, (M, N), device="cuda", generator=g).float())

    def spec():
        return dict(s=qs, S=qS, gscale=qg, b4=qb4, lo=lo, hi=hi, gelu=gelu, rowmul=rowmul, coldiv=coldiv, colmode=colmode)
    f1 = spec()
    y = ops.qgemm_i8_nt(xc, wc, bias, cs, 0.25, r, s_in, S, gscale_in, fuse=f1)
    f2 = spec()
    none = ops.qgemm_i8_nt(xc, wc, bias, cs, 0.25, r, s_in, S, gscale_in, fuse=f2, store_y=False)
    assert none is None and torch.equal(f1["codes_out"], f2["codes_out"])
    # the pair being replaced: LSQ backward on the stored y, quantiser rows = GEMM rows x heads
    geom = ops.LsqGeom(M * rowmul if not colmode else M, 1, coldiv if not colmode else N, N if not colmode else N,
                       colmode, lo, hi, 1, prologue=gelu)
    if not colmode:
        geom = ops.LsqGeom(M // T_tok, T_tok * rowmul, coldiv, N, 0, lo, hi, 1, prologue=gelu)
    geom.gscale = qg
    dx_ref, ds_ref, db4_ref, dba_ref = ops.lsq_bwd(gy, y.view(-1, geom.inner), qs, qb4, geom)
    prod = {"xcodes": xc, "wcodes": wc, "bias": bias, "w_scale": cs, "w_mult": 0.25, "r": r, "act_s": s_in, "act_S": S,
            "act_gscale": gscale_in}
    dy, ds, db4, dba = ops.qgemm_i8_lsq_bwd(gy, prod, f2)
    assert torch.equal(dy, dx_ref.view(M, N))
    assert rel_err(ds, ds_ref) < 1e-5 and rel_err(db4, db4_ref) < 1e-5 and rel_err(dba, dba_ref) < 1e-5
    # not everything was clipped (the pass-through branch of the straight-through estimator ran)
    assert float((dy == 0).float().mean()) < 0.98


@pytest.mark.gpu
@pytest.mark.parametrize("case", [
    # (name, images, heads, tokens, channels, lo, hi)
    ("three_images", 3, 2, 198, 128, -2, 1),          # 594 rows: 128-row tiles that straddle both image boundaries
    ("deit_s_block", 16, 6, 198, 384, -2, 1),         # the headline block's heads / width
    ("w4a4_256", 5, 3, 256, 256, -8, 7),              # DeiT-T at 256 px: 257 is odd -> not fusable; 256 tokens are
    ("short_last_tile", 2, 2, 130, 128, -2, 1),       # 260 rows: the third tile holds 4 rows
])
def test_fused_dqkx_recompute_backward_equals_the_two_kernels(ops, case):
    """ofq_qattn_dqkx_lsq_bwd (the qkx quantiser's backward forms its incoming gradient from dS itself) against the pair it
    replaces, ofq_qattn_dqkx_bf16s on two fp16 planes followed by ofq_qgemm_i8_lsq_bwd (attention.py:200-210, lsq.py:571-602):
    the same products in the same order, so dy and the reduced gradients are bit-identical; the maximum word of dy as well."""
    name, B, H, N, C, lo, hi = case
    if ops.GRAD_PLANES != 2:
        pytest.skip("the fused dqkx + recompute backward exists for the two-plane form (run under OFQ_GRAD_PLANES=3)")
    g = torch.Generator(device="cuda").manual_seed(len(name) + 13 * N)
    M, Nout = B * N, H * C
    ldS = (N + 15) // 16 * 16
    xc = torch.randint(-2, 2, (M, C), dtype=torch.int8, device="cuda", generator=g)
    wc = (2 * torch.randint(-2, 2, (Nout, C), device="cuda", generator=g) + 1).to(torch.int8)
    bias = torch.randn(Nout, device="cuda", generator=g) * 0.1
    cs = torch.rand(Nout, device="cuda", generator=g) * 0.05 + 0.01
    r = torch.randn(Nout, device="cuda", generator=g) * 0.3
    sx = torch.rand(N, device="cuda", generator=g) * 0.3 + 0.05
    gx = 0.013
    bax = torch.randn(C, device="cuda", generator=g) * 0.2
    qs = torch.rand(N * H, device="cuda", generator=g) * 0.3 + 0.1
    qb4 = torch.randn(Nout, device="cuda", generator=g) * 0.05
    dS = torch.randn(B, H, N, ldS, device="cuda", generator=g) * torch.pow(
        10.0, torch.randint(-4, 1, (B, H, N, ldS), device="cuda", generator=g).float())
    dS[..., N:] = float("nan")                                    # the pad columns are never read
    q = dict(s=qs, S=N * H, gscale=0.021, b4=qb4, lo=lo, hi=hi, gelu=0, rowmul=H, coldiv=C, colmode=0)
    prod = {"xcodes": xc, "wcodes": wc, "bias": bias, "w_scale": cs, "w_mult": 0.25, "r": r, "act_s": sx, "act_S": N,
            "act_gscale": gx}
    assert ops.dqkx_lsq_fusable(prod, q, sx, gx, N, C, ldS)
    ops.amax_begin(dS.device)
    try:
        gy = ops.qattn_dqkx(dS, xc, sx, gx, bax, B, H, N, C, ldS, planes=2)
        dy_ref, ds_ref, db4_ref, dba_ref = ops.qgemm_i8_lsq_bwd(gy.view(M, Nout), prod, q)
        w_ref = ops.amax_of(dy_ref)
        dy, ds, db4, dba = ops.qattn_dqkx_lsq_bwd(dS, prod, q, bax, B, H, N, C, ldS, planes=2)
        w = ops.amax_of(dy)
        assert torch.isfinite(dy_ref).all() and float((dy_ref == 0).float().mean()) < 0.98
        if C % 384 == 0:          # the pair's first kernel is the two-plane stream kernel: the same products in the same order
            assert torch.equal(dy, dy_ref)
            assert torch.equal(ds, ds_ref) and torch.equal(db4, db4_ref) and torch.equal(dba, dba_ref)
        else:                     # elsewhere ofq_qattn_dqkx_bf16s multiplies three bf16 planes: fp32-grade agreement, same gates
            assert torch.equal(dy == 0, dy_ref == 0)
            assert rel_err(dy, dy_ref) < 2e-6
            assert rel_err(ds, ds_ref) < 1e-5 and rel_err(db4, db4_ref) < 1e-5 and rel_err(dba, dba_ref) < 1e-5
        assert float(w.view(torch.float32).max()) == float(dy.abs().max())
    finally:
        ops.amax_end()


@pytest.mark.gpu
@pytest.mark.parametrize("cfg", [("qkr", 3, 6, 198, 384, 2), ("qkr_full", 128, 6, 197, 384, 2), ("qkr_small", 2, 3, 70, 96, 4), ("plain", 3, 3, 198, 64, 4),
                                 ("plain_256", 1, 2, 256, 32, 3),
                                 ("qkr_window", 64, 3, 49, 96, 3), ("plain_window", 32, 6, 49, 32, 3), ("qkr_win_tiny", 4, 2, 7, 32, 2),
                                 ("plain_win64", 6, 2, 64, 16, 4)])
def test_fused_scores_softmax_equals_the_two_kernels(ops, cfg):
    """ofq_qattn_scores_softmax_i8 (score panel kept in LDS; the *window* cases take its 64-key form) against
    ofq_qattn_scores(_plain)_i8 followed by
    ofq_softmax_lsq_fwd (attention.py:96-99 / :207-216): same expressions element for element -- the probabilities agree to
    the last bits (only the order of the row sum differs), the uint8 codes are identical except where a probability sits on a
    rounding tie, the code row sums follow the codes."""
    from ofq_amd.functional import pad16
    name, B, H, N, CK, bits = cfg
    plain = name.startswith("plain")
    g = torch.Generator(device="cuda").manual_seed(N + CK)
    lo, hi_c = -(2 ** (bits - 1)), 2 ** (bits - 1) - 1
    Np = pad16(N)
    if plain:
        C = H * CK
        ac = torch.randint(lo, hi_c + 1, (B, N, C), dtype=torch.int8, device="cuda", generator=g)
        bc = torch.randint(lo, hi_c + 1, (B, N, C), dtype=torch.int8, device="cuda", generator=g)
        sb = torch.rand(N, device="cuda", generator=g) * 0.2 + 0.05
        u = torch.randn(B * N, H, device="cuda", generator=g) * 0.1
        tq = torch.randn(B * N, H, device="cuda", generator=g) * 0.1
    else:
        C = CK
        ac = torch.randint(lo, hi_c + 1, (B, N, C), dtype=torch.int8, device="cuda", generator=g)
        bc = torch.randint(lo, hi_c + 1, (B, N, H, C), dtype=torch.int8, device="cuda", generator=g)
        sb = torch.rand(N * H, device="cuda", generator=g) * 0.2 + 0.05
        u = torch.randn(B * N, H, device="cuda", generator=g) * 0.1
        tq = torch.randn(B * N * H, device="cuda", generator=g) * 0.1
    sa = torch.rand(N, device="cuda", generator=g) * 0.2 + 0.05
    z = torch.randn(H, device="cuda", generator=g) * 0.1
    sm_s = torch.rand(N, device="cuda", generator=g) * 0.02 + 0.01
    alpha, hi = CK ** -0.5 if plain else (C // H) ** -0.5, 2 ** bits - 1
    addend = torch.randn(2, N, Np, device="cuda", generator=g) if name == "qkr_small" else None      # B*H % 2 == 0
    if "window" in name:          # Swin: relative-position bias + shift mask, one slab per (window of the image, head)
        addend = torch.randn(4 * H, N, Np, device="cuda", generator=g)
        addend[1::2, :, : N // 2] -= 100.0
    if plain:
        S = ops.qattn_scores_plain(ac, bc, sa, 0.01, sb, 0.02, u, tq, z, B, H, N, CK, Np)
    else:
        S = ops.qattn_scores(ac, bc, sa, 0.01, sb, 0.02, u, tq, z, B, H, N, C, Np)
    prob_ref, _, codes_ref, rs_ref = ops.softmax_lsq_fwd(S, sm_s, B * H * N, N, Np, N, alpha, hi, B * H * N, want_codes=True,
                                                         need_values=False, addend=addend)
    prob, codes, rs = ops.qattn_scores_softmax(ac, bc, sa, 0.01, sb, 0.02, u, tq, z, plain, sm_s, alpha, hi, B, H, N,
                                               CK, Np, addend=addend)
    assert float((prob[..., :N] - prob_ref[..., :N]).abs().max()) < 2e-7
    assert float(prob[..., N:].abs().max()) == 0.0 if Np > N else True
    differ = float((codes[..., :N] != codes_ref[..., :N]).float().mean())
    assert differ < 1e-5, differ
    assert torch.equal(rs, codes[..., :N].float().sum(-1).reshape(-1))
    assert float((prob[..., :N].sum(-1) - 1).abs().max()) < 1e-5


@pytest.mark.gpu
def test_attention_prep_launch_equals_the_three_kernels(ops):
    """ofq_qattn_prep (u, tq and the transposed v codes as three block ranges of one launch) against ofq_rowdot_i8_multi,
    ofq_rowdot_i8 and ofq_codes_transpose_i8: the same device code, so bit-identical outputs (attention.py:207-219 operands)."""
    from ofq_amd.functional import pad16
    B, H, N, C = 3, 6, 198, 384
    g = torch.Generator(device="cuda").manual_seed(5)
    xc = torch.randint(-2, 2, (B, N, C), dtype=torch.int8, device="cuda", generator=g)
    qc = torch.randint(-2, 2, (B, N, H, C), dtype=torch.int8, device="cuda", generator=g)
    vc = torch.randint(-2, 2, (B, N, C), dtype=torch.int8, device="cuda", generator=g)
    baq = torch.randn(H, C, device="cuda", generator=g)
    bax = torch.randn(C, device="cuda", generator=g)
    Np = pad16(N)
    u, tq, vT, z = ops.qattn_prep(xc, baq, qc, bax, vc, B, H, N, C, Np)
    assert rel_err(z.cpu(), (baq.double() @ bax.double()).float().cpu()) < 1e-6          # z[h] = baq[h] . bax
    assert torch.equal(u, ops.rowdot_i8_multi(xc.view(B * N, C), baq))
    assert torch.equal(tq, ops.rowdot_i8(qc.view(B * N * H, C), bax))
    assert torch.equal(vT, ops.codes_transpose_i8(vc, Np))
    assert torch.equal(vT[:, :, :N], vc.transpose(1, 2)) and int(vT[:, :, N:].abs().max()) == 0


@pytest.mark.gpu
@pytest.mark.parametrize("mode", ["plain", "mixup", "cutmix", "mixup_erase", "erase"])
def test_input_pipeline_kernel_is_bit_exact_vs_oracle(ops, mode):
    """ofq_input_pipeline_u8 (uint8-space mixup / cutmix with the mirrored sample, normalisation, random-erasing noise in one
    pass) against the oracle's restatement of timm 0.5.4's FastCollateMixup + PrefetchLoader + RandomErasing
    (train.py:579-629): byte and elementwise fp32 work, so bit for bit; the soft targets too."""
    import random
    from ofq_amd.data import DeviceInputPipeline, MixupParams, RandomErasingParams
    B, C, H, W = 8, 3, 224, 224
    g = torch.Generator().manual_seed(3)
    x = torch.randint(0, 256, (B, C, H, W), dtype=torch.uint8, generator=g)
    tgt = torch.randint(0, 1000, (B,), generator=g)
    noise = torch.randn(B, C, H, W, generator=g)
    np_seed = {"plain": 0, "mixup": 1, "cutmix": 2, "mixup_erase": 1, "erase": 0}[mode]
    for trial in range(4):
        np.random.seed(100 * np_seed + trial)
        random.seed(trial + 5)
        mix = None
        if "mix" in mode:
            mix = MixupParams(mixup_alpha=0.8 if mode != "cutmix" else 0.0, cutmix_alpha=1.0 if mode == "cutmix" else 0.0)
        er = RandomErasingParams(probability=0.6) if "erase" in mode else None
        pipe = DeviceInputPipeline(mixup=mix, erasing=er)
        out, t = pipe(x.cuda(), tgt.cuda(), noise=noise.cuda())
        last = pipe.last
        want = O.input_pipeline(x.numpy(), last["lam"], last["use_cutmix"], last["box"], last["rects"], noise.numpy())
        assert torch.equal(out.cpu(), want), (mode, trial, last)
        if mix is not None:
            assert last["lam"] != 1.0 and last["use_cutmix"] == (mode == "cutmix")
            assert torch.equal(t.cpu(), O.mixup_target(tgt, 1000, last["lam"], 0.1))
        else:
            assert torch.equal(t.cpu(), tgt)
        if er is not None and trial == 0:
            assert last["rects"] is not None and int((last["rects"][:, 2] > 0).sum()) >= 2


def test_permute_tokens_is_the_row_gather(ops):
    """ofq_permute_tokens (Swin window partition / reverse, swin.py:103-131) == index_select, and the inverse undoes it."""
    g = torch.Generator().manual_seed(5)
    for B, N, C in ((3, 3136, 96), (2, 784, 192), (5, 49, 768), (1, 7, 4)):
        x = torch.randn(B, N, C, generator=g).cuda()
        idx = torch.randperm(N, generator=g)
        inv = torch.empty_like(idx)
        inv[idx] = torch.arange(N)
        y = ops.permute_tokens(x, idx.int().cuda())
        assert torch.equal(y.cpu(), x.cpu().index_select(1, idx))
        assert torch.equal(ops.permute_tokens(y, inv.int().cuda()), x)
    with pytest.raises(ValueError):
        ops.permute_tokens(x, idx.cuda())                      # int64 index


@pytest.mark.parametrize("shape", [(37, 3, 49, 32, 64), (5, 12, 49, 32, 64), (3, 24, 49, 32, 64), (6, 2, 64, 16, 64),
                                   (9, 4, 33, 64, 48), (7, 1, 16, 16, 16),
                                   (2, 6, 198, 64, 208), (3, 3, 198, 64, 208), (2, 2, 100, 32, 112), (2, 1, 65, 16, 80)])
def test_window_sized_attention_backward_kernels_vs_fp64(ops, shape):
    """Swin-sized (window, head) batches run on the one-wave-per-tile kernel (qgemm_bf16s_tn_win_kernel): dV, dqkx
    (several 64-column blocks, chunked past 384 columns) and the plain-attention dk, against fp64; the pad columns of dS
    hold NaN (they only ever meet output rows that are not stored).  The 198-token shapes take the workgroup-tile kernels (same entry points)."""
    B, H, N, d, Np = shape
    C = H * d
    rs = np.random.RandomState(11)
    gx, gp, gq = 0.011, 0.017, 0.013
    cu = lambda t: t.cuda()
    xcodes = torch.from_numpy(rs.randint(-4, 4, (B, N, C)).astype(np.int8))
    pcodes = torch.zeros(B, H, N, Np, dtype=torch.uint8)
    pcodes[..., :N] = torch.from_numpy(rs.randint(0, 8, (B, H, N, N)).astype(np.uint8))
    sx, sp = T(det_uniform((N,), 201, 0.2, 1.0)), T(det_uniform((N,), 203, 0.05, 0.3))
    bax = T(det_uniform((C,), 205, -0.1, 0.1))
    ax, ap = O.lsq_effective_scale(sx, gx).double(), O.lsq_effective_scale(sp, gp).double()
    dO = T(det_normalish((B, N, C), 208, 1.0))
    ph = ap[None, None, :, None] * pcodes[..., :N].double()
    dV = ops.qattn_dv(cu(dO), cu(pcodes).view(torch.int8), cu(sp), gp, B, H, N, d, Np)
    dV_ref = torch.einsum("bhnm,bnhj->bmhj", ph, dO.double().view(B, N, H, d)).reshape(B, N, C)
    assert rel_err(dV.cpu(), dV_ref.float()) < 2e-6
    # dP on wave tiles (qgemm_bf16s_nt_win_kernel): per-channel step of v, per-row addend w
    gv = 0.019
    vcodes = torch.from_numpy(rs.randint(-8, 8, (B, N, C)).astype(np.int8))
    sv, bav = T(det_uniform((C,), 204, 0.2, 1.0)), T(det_uniform((C,), 207, -0.1, 0.1))
    av = O.lsq_effective_scale(sv, gv).double()
    w = ops.rowdot_f32_seg(cu(dO).view(B * N, C), cu(bav), H, d)
    dP = ops.qattn_dp(cu(dO), cu(vcodes), cu(sv), gv, w, B, H, N, d, Np)
    vh = av * vcodes.double() + bav.double()
    dP_ref = torch.einsum("bnhj,bmhj->bhnm", dO.double().view(B, N, H, d), vh.view(B, N, H, d))
    assert rel_err(dP.cpu()[..., :N], dP_ref.float()) < 2e-6
    dS = torch.full((B, H, N, Np), float("nan"))
    dS[..., :N] = T(det_normalish((B, H, N, N), 209, 1.0))
    xh = ax[None, :, None] * xcodes.double() + bax.double()
    dq = ops.qattn_dqkx(cu(dS), cu(xcodes), cu(sx), gx, cu(bax), B, H, N, C, Np)
    dq_ref = torch.einsum("bhnm,bnc->bmhc", dS[..., :N].double(), xh)
    assert rel_err(dq.cpu(), dq_ref.float()) < 2e-6
    dq0 = ops.qattn_dqkx(cu(dS), cu(xcodes), cu(sx), gx, None, B, H, N, C, Np)            # no offset term
    assert rel_err(dq0.cpu(), torch.einsum("bhnm,bnc->bmhc", dS[..., :N].double(), ax[None, :, None] * xcodes.double()).float()) < 2e-6
    if d % 16 == 0:
        sq, bq = T(det_uniform((N,), 211, 0.2, 1.0)), T(det_uniform((C,), 212, -0.1, 0.1))
        aq = O.lsq_effective_scale(sq, gq).double()
        qh = (aq[None, :, None] * xcodes.double() + bq.double()).view(B, N, H, d)
        dk = ops.qattn_dk_plain(cu(dS), cu(xcodes), cu(sq), gq, cu(bq), B, H, N, d, Np)
        dk_ref = torch.einsum("bhnm,bnhj->bmhj", dS[..., :N].double(), qh).reshape(B, N, C)
        assert rel_err(dk.cpu(), dk_ref.float()) < 2e-6



@pytest.mark.parametrize("BK", [(128, 1000), (7, 10), (33, 257)])
def test_kd_loss_kernel_equals_the_torch_ops(ops, BK):
    """ofq_kd_loss_fwd / _bwd (KDLossSoftandHard, src/quantization/utils.py:59-77; train.py:906-913) against the stock ops the
    reference uses: loss value and both logit gradients, also under a non-unit upstream gradient."""
    import torch.nn.functional as F
    from ofq_amd.functional import KDLossFn, kd_loss_fusable
    B, K = BK
    g = torch.Generator(device="cuda").manual_seed(5)
    c = (torch.randn(B, K, device="cuda", generator=g) * 3).requires_grad_(True)
    d = (torch.randn(B, K, device="cuda", generator=g) * 3).requires_grad_(True)
    t = torch.randn(B, K, device="cuda", generator=g) * 2
    y = torch.randint(0, K, (B,), device="cuda", generator=g)
    assert kd_loss_fusable(c, d, t, y)
    ref = -torch.sum(F.softmax(t, dim=1) * F.log_softmax(d, dim=1), dim=1).mean() + F.cross_entropy(c, y)
    (ref * 1.7).backward()
    gc, gd = c.grad.clone(), d.grad.clone()
    c.grad = d.grad = None
    loss = KDLossFn.apply(c, d, t, y)
    (loss * 1.7).backward()
    assert abs(float(loss) - float(ref)) < 2e-6 * abs(float(ref))
    assert rel_err(c.grad.cpu(), gc.cpu()) < 2e-6 and rel_err(d.grad.cpu(), gd.cpu()) < 2e-6
    # through the module the reference's recipes construct
    from ofq_amd.quantization.utils import KDLossSoftandHard
    assert abs(float(KDLossSoftandHard()((c.detach(), d.detach()), y, t)) - float(ref)) < 2e-6 * abs(float(ref))
    # nn.CrossEntropyLoss's ignore_index (-100): such rows give neither loss nor gradient, the hard term averages over the others
    yi = y.clone()
    yi[::3] = -100
    c.grad = d.grad = None
    ref = -torch.sum(F.softmax(t, dim=1) * F.log_softmax(d, dim=1), dim=1).mean() + F.cross_entropy(c, yi)
    (ref * 0.6).backward()
    gc, gd = c.grad.clone(), d.grad.clone()
    c.grad = d.grad = None
    loss = KDLossFn.apply(c, d, t, yi)
    (loss * 0.6).backward()
    assert abs(float(loss) - float(ref)) < 2e-6 * abs(float(ref))
    assert rel_err(c.grad.cpu(), gc.cpu()) < 2e-6 and rel_err(d.grad.cpu(), gd.cpu()) < 2e-6
    assert float(c.grad[::3].abs().max()) == 0.0
    # a label outside [0, K) that is not ignore_index (the stock op traps): the loss is NaN, not a silently different number
    yb = y.clone()
    yb[0] = K
    assert torch.isnan(KDLossFn.apply(c.detach(), d.detach(), t, yb))


@pytest.mark.parametrize("dist", [True, False])
def test_token_assembly_kernel_is_cat_plus_pos_embed(ops, dist):
    """ofq_assemble_tokens (deit.py:32-44): values bit for bit, gradients against autograd of the cat + add."""
    from ofq_amd.functional import AssembleTokensFn
    B, P, C = 5, 196, 192
    g = torch.Generator(device="cuda").manual_seed(6)
    mk = lambda *s: torch.randn(*s, device="cuda", generator=g).requires_grad_(True)      # noqa: E731
    x, cls, pos = mk(B, P, C), mk(1, 1, C), mk(1, P + (2 if dist else 1), C)
    dtk = mk(1, 1, C) if dist else None
    parts = [cls.expand(B, -1, -1)] + ([dtk.expand(B, -1, -1)] if dist else []) + [x]
    ref = torch.cat(parts, dim=1) + pos
    up = torch.randn(ref.shape, device="cuda", generator=g)
    (ref * up).sum().backward()
    want = [t.grad.clone() for t in (x, cls, pos)] + ([dtk.grad.clone()] if dist else [])
    for t in (x, cls, pos, dtk):
        if t is not None:
            t.grad = None
    out = AssembleTokensFn.apply(x, cls, dtk, pos)
    assert torch.equal(out, ref)
    (out * up).sum().backward()
    got = [t.grad for t in (x, cls, pos)] + ([dtk.grad] if dist else [])
    assert torch.equal(got[0], want[0])
    for a, b in zip(got[1:], want[1:]):
        assert a.shape == b.shape and rel_err(a.cpu(), b.cpu()) < 2e-6


@pytest.mark.gpu
@pytest.mark.parametrize("case", [("deit_s", 3, 6, 198), ("deit_t_197", 2, 3, 197), ("short", 2, 2, 33), ("full_224", 1, 2, 224),
                                  ("one_key", 1, 1, 1)])
def test_teacher_attention_in_one_launch_vs_fp64(ops, case):
    """ofq_attn_f32_fwd (the KD teacher's softmax(scale q k^T) v, deit_vision_transformer.py:85-116 without quantisers, on fp16
    planes with tile-local scales) against the same attention in fp64, and against the fp32 computation's own distance from
    it: fp32-grade means within a small multiple of what plain fp32 loses.  Inputs with outliers (a few keys 30x the rest: a
    peaked softmax) and columns spanning decades (the plane scale is per tile, not per column)."""
    name, B, H, N = case
    d, C = 64, H * 64
    g = torch.Generator(device="cuda").manual_seed(N * 7 + H)
    qkv = torch.randn(B * N, 3 * C, device="cuda", generator=g)
    qkv[:, :C] *= 2.0
    qkv[::17, C:2 * C] *= 30.0                                        # outlier keys
    qkv[:, 2 * C:] *= torch.pow(10.0, torch.randint(-3, 2, (3 * C - 2 * C,), device="cuda", generator=g).float())
    scale = d ** -0.5
    out = ops.attn_f32_fwd(qkv, B, H, N, d, scale)

    def ref(x):
        q, k, v = (x[:, i * C:(i + 1) * C].view(B, N, H, d).permute(0, 2, 1, 3) for i in range(3))
        a = torch.softmax((q @ k.transpose(-1, -2)) * scale, dim=-1)
        return (a @ v).permute(0, 2, 1, 3).reshape(B * N, C)
    r64 = ref(qkv.double())
    r32 = ref(qkv)
    assert torch.isfinite(out).all()
    # per output column (the v columns span decades): error against the column's largest value
    colmax = r64.abs().amax(dim=0).clamp_min(1e-30)
    e_hip = ((out.double() - r64).abs() / colmax).max().item()
    e_f32 = ((r32.double() - r64).abs() / colmax).max().item()
    assert e_hip < 3 * e_f32 + 1e-6, (e_hip, e_f32)


@pytest.mark.gpu
@pytest.mark.parametrize("mnk", [(25344, 1152, 384), (25344, 384, 1536), (1000, 1536, 384), (4096, 384, 96)])
def test_teacher_linear_on_fp16_planes_three_and_four_products(ops, mnk):
    """ops.linear_f16x4 (the KD teacher's fp32 linear layers: the weight split once into two fp16 planes, the activations
    in the kernel; ofq_qgemm_bf16s_nt_sk with a column bias) with four plane products and with three (ofq_nt_seg.hi_only: the
    trailing planes' product skipped) against the fp64 product: both within a small multiple of what a plain fp32 GEMM loses."""
    M, N, K = mnk
    g = torch.Generator(device="cuda").manual_seed(M + N + K)
    x = torch.randn(M, K, device="cuda", generator=g) * torch.pow(10.0, torch.randint(-2, 2, (K,), device="cuda", generator=g).float())
    W = torch.randn(N, K, device="cuda", generator=g) * 0.05
    b = torch.randn(N, device="cuda", generator=g)
    planes = ops.split_f32_f16x2(W)
    ref = x.double() @ W.double().t() + b.double()
    f32 = (x @ W.t() + b).double()
    scale = ref.abs().max().item()
    e32 = (f32 - ref).abs().max().item() / scale
    for products in (4, 3):
        ops.amax_begin(x.device)
        try:
            ops.absmax(x)
            y = ops.linear_f16x4(x, planes, b, products=products)
        finally:
            ops.amax_end()
        e = (y.double() - ref).abs().max().item() / scale
        assert e < 3 * e32 + 2e-7, (products, e, e32)
