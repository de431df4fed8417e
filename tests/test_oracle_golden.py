"""Pin oracle/ofq_oracle.py against golden vectors produced by the reference itself
(tests/golden/make_golden.py, run in the build container against /root/reference).  CPU only."""
import os

import numpy as np
import pytest
import torch

import ofq_oracle as O
from detgen import det_uniform, det_normalish
from util import load_golden, group, case_names, params, T, rel_err

TOL = 1e-6   # oracle vs reference on the same CPU: same ops, same order


def test_g1_statsq_bit_exact():
    d = load_golden("g1_statsq")
    for c in range(int(d["ncases"])):
        g = group(d, "c%d" % c)
        bits = int(g["shape"][2])
        W = T(g["W"]).requires_grad_(True)
        y, L, s = O.statsq(W, bits)
        (y * T(g["g"])).sum().backward()
        assert torch.equal(y.detach(), T(g["y"]))
        assert torch.equal(L.to(torch.int8), T(g["L"]))
        assert torch.equal(s.squeeze(), T(g["s"]))
        assert torch.equal(W.grad, T(g["dW"]))
        n = 2 ** (bits - 1)
        assert int(L.min()) >= -n and int(L.max()) <= n - 1


def _run_lsq(name, g):
    x = T(g["x"]).requires_grad_(True)
    s = T(g["s"]).requires_grad_(True)
    lo, hi = int(g["lohi"][0]), int(g["lohi"][1])
    unsigned = lo == 0
    bits = int(round(np.log2(hi - lo + 1)))
    if name.startswith("token"):
        # token3d cases had edge values planted into token 0 AFTER the init pass (make_golden.plant_token)
        k0 = 1 if name.startswith("token3d") else 0
        assert torch.allclose(O.lsq_token_init(x, bits, unsigned)[k0:], T(g["s_init"])[k0:], rtol=1e-6, atol=0)
        y = O.lsq_token(x, s, bits, unsigned)
        alpha = s.detach().unsqueeze(-1)
    elif name.startswith("chan"):
        assert torch.allclose(O.lsq_channel_init(x, bits), T(g["s_init"]), rtol=1e-6, atol=0)
        y = O.lsq_channel(x, s, bits)
        alpha = s.detach()
    elif name.startswith("img"):
        signed = bool(g["signed"][0] != 0)
        assert signed == (lo < 0)
        assert torch.allclose(O.lsq_img_init(x, signed), T(g["s_init"]), rtol=1e-6, atol=0)
        y = O.lsq_img(x, s, signed)
        alpha = s.detach().view(1, -1, 1, 1)
    elif name == "convw":
        assert torch.allclose(O.lsq_convw_init(x), T(g["s_init"]), rtol=1e-6, atol=0)
        y = O.lsq_convw(x, s)
        alpha = s.detach().view(-1, 1, 1, 1)
    elif name == "roww":
        assert torch.allclose(O.lsq_roww_init(x), T(g["s_init"]), rtol=1e-6, atol=0)
        y = O.lsq_roww(x, s)
        alpha = s.detach().unsqueeze(-1)
    elif name == "tensor":
        assert torch.allclose(O.lsq_tensor_init(x), T(g["s_init"]), rtol=1e-6, atol=0)
        y = O.lsq_tensor(x, s)
        alpha = s.detach()
    else:
        raise AssertionError(name)
    gy = T(g["g"])
    (y * gy).sum().backward()
    assert torch.equal(y.detach(), T(g["y"])), name
    assert torch.equal(x.grad, T(g["dx"])), name
    assert rel_err(s.grad, g["ds"]) < TOL, name
    # closed form == autograd (this is the formula the HIP backward kernel implements)
    M = x.numel() // s.numel()
    dx_cf, da_cf = O.lsq_backward_closed_form(gy, x.detach(), alpha, lo, hi, 1.0 / np.sqrt(hi * M))
    assert torch.equal(dx_cf, T(g["dx"])), name
    red = da_cf.double()
    # reduce over every broadcast axis of alpha
    while red.dim() > alpha.dim():
        red = red.sum(0)
    for ax in range(alpha.dim()):
        if alpha.shape[ax] == 1 and red.shape[ax] != 1:
            red = red.sum(ax, keepdim=True)
    assert rel_err(red.reshape(-1), g["ds"].reshape(-1)) < 1e-5, name
    # integer levels stay inside [lo, hi]
    q = O.lsq_levels(x.detach(), alpha, lo, hi, 1.0 / np.sqrt(hi * M))
    assert int(q.min()) >= lo and int(q.max()) <= hi


def test_g2_lsq_all_variants():
    d = load_golden("g2_lsq")
    names = case_names(d)
    assert len(names) >= 25
    for nme in names:
        kind = nme.split("_b")[0] if "_b" in nme else nme
        _run_lsq(kind if kind in ("convw", "roww", "tensor") else nme, group(d, nme))


def _check_module(g, fwd, tol=TOL):
    p = params(g)
    x = g["_x"].clone().requires_grad_(True)
    y = fwd(x, p)
    (y * T(g["g"])).sum().backward()
    assert rel_err(y.detach(), g["y"]) < tol
    assert rel_err(x.grad, g["dx"]) < tol
    for k, v in g.items():
        if k.startswith("grad:"):
            assert p[k[5:]].grad is not None, k
            assert rel_err(p[k[5:]].grad, v) < 10 * tol, k


def test_g3_qlinear():
    d = load_golden("g3_qlinear")
    for nme in case_names(d):
        g = group(d, nme)
        B, N, I, Oo, wb, ab, sym, seed = [int(v) for v in g["meta"]]
        x = T(det_normalish((B, N, I), seed, 1.0))
        if not sym:
            x = x.abs()
        g["_x"] = x
        _check_module(g, lambda x, p: O.qlinear(x, p, wb, ab, unsigned=not sym))


def test_g5_qmlp():
    d = load_golden("g5_qmlp")
    for nme in case_names(d):
        g = group(d, nme)
        B, N, C, Hd, wb, ab, seed = [int(v) for v in g["meta"]]
        g["_x"] = T(det_normalish((B, N, C), seed, 1.0))
        _check_module(g, lambda x, p: O.qmlp(x, p, wb, ab))


def test_g4_attention_plain_and_qkr():
    d = load_golden("g4_attention")
    names = case_names(d)
    assert any(n.startswith("plain") for n in names) and any(n.startswith("qkr_") for n in names)
    for nme in names:
        g = group(d, nme)
        B, N, C, H, wb, ab, seed = [int(v) for v in g["meta"]]
        g["_x"] = T(det_normalish((B, N, C), seed, 1.0))
        fn = O.qattention if nme.startswith("plain") else O.qattention_qkr
        _check_module(g, lambda x, p: fn(x, p, H, wb, ab), tol=2e-6)


def test_g6_stem_and_head():
    d = load_golden("g6_stem_head")
    for nme in ("conv_signed", "conv_unsigned"):
        g = group(d, nme)
        lo, hi = float(g["img_lohi"][0]), float(g["img_lohi"][1])
        g["_x"] = T(det_uniform((1, 3, 224, 224), int(g["meta"][2]), lo, hi))
        p = params(g)
        x = g["_x"].clone().requires_grad_(True)
        y = O.qconv_patch_embed(x, p, 16)                      # (B, 196, C)
        yref = T(g["y"]).flatten(2).transpose(1, 2)
        gy = T(g["g"]).flatten(2).transpose(1, 2)
        (y * gy).sum().backward()
        assert rel_err(y.detach(), yref) < TOL
        assert rel_err(x.grad, g["dx"]) < TOL
        for k, v in g.items():
            if k.startswith("grad:"):
                assert rel_err(p[k[5:]].grad, v) < 1e-5, k
    g = group(d, "head")
    B, I, Oo, seed = [int(v) for v in g["meta"]]
    g["_x"] = T(det_normalish((B, I), seed, 1.0))
    _check_module(g, lambda x, p: O.qhead(x, p))


def test_g7_tiny_deit_full_step():
    d = load_golden("g7_tiny_deit")
    for nme in case_names(d):
        g = group(d, nme)
        B, depth, dim, heads, wb, ab, qkr, seed, ncls, mlp_ratio = [int(v) for v in g["meta"]]
        cfg = dict(depth=depth, num_heads=heads, patch=16, wbits=wb, abits=ab, qkr=bool(qkr))
        p = params(g)
        img = T(det_uniform((B, 3, 224, 224), seed, -2.0, 2.0))
        cls_o, dist_o = O.deit_forward(img, p, cfg, training=True)
        loss = O.kd_loss_soft_and_hard(cls_o, dist_o, T(g["target"]), T(g["soft"]))
        loss.backward()
        assert rel_err(cls_o.detach(), g["cls"]) < 1e-5
        assert rel_err(dist_o.detach(), g["dist"]) < 1e-5
        assert abs(float(loss) - float(g["loss"])) < 1e-5 * abs(float(g["loss"]))
        n_checked = 0
        for k, v in g.items():
            if k.startswith("grad:"):
                gp = p[k[5:]].grad
                assert gp is not None, k
                assert rel_err(gp, v) < 1e-4, k
                n_checked += 1
        assert n_checked > 60
        with torch.no_grad():
            ev = O.deit_forward(img, p, cfg, training=False)
        assert rel_err(ev, g["eval_logits"]) < 1e-5


def test_g8_cga():
    d = load_golden("g8_cga")
    for c in range(int(d["ncases"])):
        g = group(d, "c%d" % c)
        bits = int(g["meta"][2])
        br = float(g["br"])
        W = T(g["W"])
        frz = O.cga_freeze_idx(W, bits, br)
        assert torch.equal(frz, T(g["frz"]))
        assert 0.0 < float(frz.mean()) < 1.0
        gm = O.cga_mask_grad(T(g["g"]), frz)
        assert torch.equal(gm, T(g["gm"]))
        # idempotence of the mask (property from SURVEY.md §4)
        assert torch.equal(O.cga_mask_grad(gm, frz), gm)
        Wp = torch.nn.Parameter(W.clone())
        opt = torch.optim.AdamW([Wp], lr=1e-3, weight_decay=0.05)
        Wp.grad = gm.clone()
        opt.step()
        Wn = O.cga_restore(Wp.detach(), W, frz)
        assert torch.equal(Wn, T(g["W_after"]))
        assert torch.equal(Wn[frz == 1], W[frz == 1])


def test_statsq_scale_invariance_property():
    # Q(aW) = a Q(W) for a power-of-two a (exact in fp32); levels unchanged (SURVEY.md §4 'property')
    W = T(det_normalish((16, 64), 77, 0.02))
    for bits in (2, 3, 4):
        y1, L1, _ = O.statsq(W, bits)
        y2, L2, _ = O.statsq(W * 4.0, bits)
        assert torch.equal(L1, L2)
        assert torch.equal(y1 * 4.0, y2)


def test_g9_swin_modules():
    d = load_golden("g9_swin_modules")
    names = case_names(d)
    assert sum(n.startswith("attn_") for n in names) >= 9
    for nme in names:
        g = group(d, nme)
        if nme.startswith("attn_"):
            B, Hh, Ww, C, H, wb, ab, shift, seed = [int(v) for v in g["meta"]]
            g["_x"] = T(det_normalish((B, Hh, Ww, C), seed, 1.0))
            qkr = not nme.startswith("attn_plain")
            _check_module(g, lambda x, p: O.swin_window_attention(x, p, H, [7, 7], [shift, shift], wb, ab, qkr), tol=5e-6)
        elif nme == "reduction4d":
            B, Hh, Ww, I, Oo, wb, ab, seed = [int(v) for v in g["meta"]]
            g["_x"] = T(det_normalish((B, Hh, Ww, I), seed, 1.0))
            _check_module(g, lambda x, p: O.qlinear(x, p, wb, ab))
            assert g["p:input_quant_fn.s"].shape == (Ww,)                 # step indexed by the feature-map column
        else:
            B, Hh, Ww, C, Hd, wb, ab, seed = [int(v) for v in g["meta"]]
            g["_x"] = T(det_normalish((B, Hh, Ww, C), seed, 1.0))
            _check_module(g, lambda x, p: O.qmlp(x, p, wb, ab))


def test_g9_swin_tiny_full_step():
    d = load_golden("g9_swin_tiny")
    for nme in case_names(d):
        g = group(d, nme)
        meta = [int(v) for v in g["meta"]]
        B, dim, wb, ab, qkr, seed, ncls = meta[:7]
        depths, heads = meta[7:9], meta[9:11]
        cfg = dict(depths=depths, num_heads=heads, window=[7, 7], patch=4, wbits=wb, abits=ab, qkr=bool(qkr))
        p = params(g)
        img = T(det_uniform((B, 3, 224, 224), seed, -2.0, 2.0))
        logits = O.swin_forward(img, p, cfg)
        loss = O.kd_loss_soft_and_hard(logits, logits, T(g["target"]), T(g["soft"]))
        loss.backward()
        assert rel_err(logits.detach(), g["logits"]) < 1e-5
        assert abs(float(loss.detach()) - float(g["loss"])) < 1e-5 * abs(float(g["loss"]))
        n = 0
        for k, v in g.items():
            if k.startswith("grad:"):
                gp = p[k[5:]].grad
                assert gp is not None, k
                den = np.abs(v).max()
                if den < 1e-6:
                    continue
                assert rel_err(gp, v) < 2e-4, k
                n += 1
        assert n > 80


def _ofq_namespace():
    from ofq_amd.deit_vision_transformer import Attention, Mlp
    from ofq_amd.swin import ShiftedWindowAttention, MLP
    from ofq_amd.quantization.modules.qlinear import QLinear, QMLP
    from ofq_amd.quantization.modules.attention import QAttention, QAttention_qkreparam
    from ofq_amd.quantization.modules.swin_attention_and_mlp import QAttention_swin_qkreparam, QMLP_swin
    return {"QLinear": QLinear, "QMLP": QMLP, "Mlp": Mlp, "Attention": Attention, "QAttention": QAttention,
            "QAttention_qkreparam": QAttention_qkreparam, "ShiftedWindowAttention": ShiftedWindowAttention,
            "QAttention_swin_qkreparam": QAttention_swin_qkreparam, "QMLP_swin": QMLP_swin, "swin_MLP": MLP}


def prod_oracle_forward(name):
    """The oracle function of a tests/golden/prodcases.py case as f(x, params)."""
    import prodcases as PC
    c = PC.CASES[name]
    k, wb, ab = c["kind"], c["wb"], c["ab"]
    if k == "qlinear":
        return lambda x, p: O.qlinear(x, p, wb, ab, unsigned=not c["sym"])
    if k in ("qmlp", "swin_mlp"):
        return lambda x, p: O.qmlp(x, p, wb, ab)
    if k == "qkr":
        return lambda x, p: O.qattention_qkr(x, p, c["H"], wb, ab)
    if k == "plain":
        return lambda x, p: O.qattention(x, p, c["H"], wb, ab)
    return lambda x, p: O.swin_window_attention(x, p, c["H"], [7, 7], [c["shift"], c["shift"]], wb, ab, True)


def prod_params(name, g):
    """Parameter dict of a production-dimension case: the big weights come from the seeded constructor (the drop-in
    module's own: its q / k / v split and copies are part of what is compared), everything else from the fixture."""
    import prodcases as PC
    q, x, _ = PC.build(name, _ofq_namespace())
    p = {k: v.detach().clone() for k, v in q.state_dict().items()}
    for k, v in g.items():
        if k.startswith("p:"):
            p[k[2:]] = T(v).clone()
        if k.startswith("w2:"):
            assert abs(float(p[k[3:]].double().norm()) - float(v)) < 1e-9 * float(v), k     # same weights as the generator's
    for k, v in p.items():
        if v.dtype.is_floating_point and "clip_val" not in k and "relative_position_index" not in k:
            v.requires_grad_(True)
    return p, x


def test_g10_production_dimension_modules():
    """The oracle against the reference at the dimensions of the production kernels (DeiT-S fc1 / fc2 / QMLP / QKR
    attention C=384 H=6, DeiT-T plain attention C=192 H=3, Swin-T window attention dim 96 and dim 384, QMLP_swin at
    28 x 28 x 192): outputs, input gradient and every parameter gradient."""
    import prodcases as PC
    d = load_golden("g10_prod")
    assert set(case_names(d)) == set(PC.CASES)
    for name in PC.CASES:
        g = group(d, name)
        p, x = prod_params(name, g)
        x = x.clone().requires_grad_(True)
        y = prod_oracle_forward(name)(x, p)
        (y * PC.upstream_grad(name, y.shape)).sum().backward()
        errs = {"y": PC.compare(y, g, "y"), "dx": PC.compare(x.grad, g, "dx")}
        n = 0
        for k in list(g):
            if k.startswith("grad:"):
                pn = k[5:].split("@")[0]
                if pn in errs:
                    continue
                assert p[pn].grad is not None, (name, pn)
                errs[pn] = PC.compare(p[pn].grad, g, "grad:" + pn)
                n += 1
        assert n >= 5, (name, n)
        for k, e in errs.items():
            # summation order (einsum / bmm blocking) is the only difference: 1e-5 on the tensor scale
            assert e["max"] < 2e-5 and e["l2"] < 2e-5 and e.get("norm", 0.0) < 1e-5, (name, k, e)


def test_g11_fp32_teacher():
    """The oracle's fp32 distilled DeiT (the KD teacher, train.py:428-442) against the reference's unquantised
    DistilledVisionTransformer: train-mode (cls, dist) and eval-mode averaged logits."""
    from ofq_amd.deit import DistilledVisionTransformer
    from functools import partial
    import torch.nn as nn
    d = load_golden("g11_teacher")
    for name in case_names(d):
        g = group(d, name)
        dim, depth, heads, B, ncls, seed = [int(v) for v in g["meta"]]
        model = DistilledVisionTransformer(img_size=224, patch_size=16, embed_dim=dim, depth=depth, num_heads=heads,
                                           mlp_ratio=4, qkv_bias=True, num_classes=ncls,
                                           norm_layer=partial(nn.LayerNorm, eps=1e-6), act_layer=nn.GELU)
        teacher_fill(model, seed)
        w2 = float(sum(p.double().pow(2).sum() for p in model.parameters()) ** 0.5)
        assert abs(w2 - float(g["w2"])) < 1e-9 * w2
        sd = {k: v.detach() for k, v in model.state_dict().items()}
        img = T(det_uniform((B, 3, 224, 224), seed + 900, -2.0, 2.0))
        with torch.no_grad():
            c, dd = O.deit_fp32_forward(img, sd, depth, heads, training=True)
            ev = O.deit_fp32_forward(img, sd, depth, heads, training=False)
        assert rel_err(c, g["cls"]) < 1e-5 and rel_err(dd, g["dist"]) < 1e-5 and rel_err(ev, g["eval"]) < 1e-5


def teacher_fill(model, seed):
    """make_golden.teacher_fill: parameters from detgen seeds by parameter index."""
    with torch.no_grad():
        for i, (n, p) in enumerate(model.named_parameters()):
            if p.dim() >= 2 and "norm" not in n:
                p.copy_(T(det_normalish(tuple(p.shape), seed + i, 0.05)))
            elif "norm" in n and n.endswith("weight"):
                p.copy_(T(det_uniform(tuple(p.shape), seed + i, 0.8, 1.2)))
            else:
                p.copy_(T(det_uniform(tuple(p.shape), seed + i, -0.1, 0.1)))


@pytest.mark.skipif(not os.path.isdir("/root/reference/src"), reason="the reference tree only exists in the build container")
def test_committed_fixtures_are_what_the_generator_writes(tmp_path):
    """tests/golden/make_golden.py (which imports the reference itself) regenerates every committed fixture bit for bit:
    each generator seeds itself, so any subset in any order writes the same files."""
    import subprocess
    import sys
    import numpy as np
    here = os.path.dirname(os.path.abspath(__file__))
    env = dict(os.environ, OFQ_GOLDEN_OUT=str(tmp_path))
    # reversed order on purpose: the files must not depend on which generators ran before
    order = ["g11", "g10", "g9", "g8", "g7", "g6", "g5", "g4", "g3", "g2", "g1"]
    subprocess.check_call([sys.executable, os.path.join(here, "golden", "make_golden.py")] + order, env=env,
                          stdout=subprocess.DEVNULL)
    committed = sorted(f for f in os.listdir(os.path.join(here, "golden")) if f.endswith(".npz"))
    assert committed == sorted(f for f in os.listdir(tmp_path) if f.endswith(".npz"))
    for f in committed:
        a, b = np.load(os.path.join(here, "golden", f)), np.load(os.path.join(tmp_path, f))
        assert sorted(a.files) == sorted(b.files), f
        for k in a.files:
            assert a[k].shape == b[k].shape and np.array_equal(a[k], b[k]), (f, k)
