"""Production-dimension module cases shared by the golden generator (reference classes, make_golden.py g10) and the
parity tests (ofq_amd classes on the GPU, oracle functions on the CPU).

Every case builds the fp32 source module with weights from detgen seeds, wraps it in the quantised class under test and
returns the input.  The weights are NOT stored in the fixture (they are regenerable, and the Q-module constructors copy /
split them exactly as the reference's do, which the comparison therefore covers as well); what the fixture stores is what
the reference computed: the lazily initialised LSQ steps, outputs and gradients -- large tensors as a strided sample plus
norms (`compact`), see `compare`.

Shapes are the ones that select the production kernel instances (VERDICT r2 "missing" item 1):
DeiT-S  fc1 384->1536 / fc2 1536->384 (qlinear.py:58-73), QMLP (qlinear.py:123-136), QAttention_qkreparam C=384 H=6
(attention.py:174-222), DeiT-T QAttention C=192 H=3 (attention.py:67-105), Swin-T window attention dim 96 / 3 heads and
dim 384 / 12 heads with shift (swin_attention_and_mlp.py:253-461), QMLP_swin at 28 x 28 x 192 (:24-63).
"""
import numpy as np
import torch
import torch.nn as nn

from detgen import det_uniform, det_normalish

STRIDE = 17            # sampling stride of tensors stored in compact form
FULL_MAX = 40_000      # tensors up to this many elements are stored whole


def T(a):
    return torch.from_numpy(np.ascontiguousarray(a))


def _fill(p, seed, std=None, lo=None, hi=None):
    with torch.no_grad():
        if std is not None:
            p.copy_(T(det_normalish(tuple(p.shape), seed, std)))
        else:
            p.copy_(T(det_uniform(tuple(p.shape), seed, lo, hi)))


# name: (kind, dims..., wbits, abits, seed).  Seeds are chosen TIE-FREE: the generator re-runs the reference in fp64 and with
# one thread and requires every output / gradient to agree with the 4-thread fp32 run to 1e-5 (make_golden.g10_prod) --
# a fake-quantised module is discontinuous, and a case in which some value sits within fp32 rounding noise of a rounding tie
# has no single right answer (seed 3060 of swin_qkr_384 was such a case: fp32 vs fp64 of the reference itself differ by 3e-2,
# and so do two CPUs' fp32 runs).
CASES = {
    "qlin_fc1": dict(kind="qlinear", B=2, N=198, I=384, O=1536, wb=2, ab=2, sym=True, seed=3000),
    "qlin_fc2": dict(kind="qlinear", B=2, N=198, I=1536, O=384, wb=2, ab=2, sym=False, seed=3010),
    "qmlp_s": dict(kind="qmlp", B=2, N=198, C=384, Hd=1536, wb=2, ab=2, seed=3020),
    "qkr_s": dict(kind="qkr", B=2, N=198, C=384, H=6, wb=2, ab=2, seed=3030),
    "plain_t": dict(kind="plain", B=2, N=198, C=192, H=3, wb=4, ab=4, seed=3040),
    "swin_qkr_96": dict(kind="swin_qkr", B=2, Hh=56, Ww=56, C=96, H=3, shift=3, wb=3, ab=3, seed=3050),
    "swin_qkr_384": dict(kind="swin_qkr", B=2, Hh=14, Ww=14, C=384, H=12, shift=3, wb=3, ab=3, seed=23060),
    "swin_mlp_192": dict(kind="swin_mlp", B=2, Hh=28, Ww=28, C=192, Hd=768, wb=3, ab=3, seed=3070),
}


def build(name, ns):
    """ns: dict of classes (reference's or ofq_amd's).  Returns (q_module, x, out_index)."""
    c = CASES[name]
    k, sd = c["kind"], c["seed"]
    # the attention constructors keep freshly initialised proj (and qkv) weights -- a reference quirk, see
    # ofq_amd/quantization/modules/attention.py -- so the torch generator is part of the case definition: both sides must
    # draw the same numbers in the same order
    torch.manual_seed(sd)
    if k == "qlinear":
        m = nn.Linear(c["I"], c["O"])
        _fill(m.weight, sd, std=0.05)
        _fill(m.bias, sd + 1, lo=-0.1, hi=0.1)
        q = ns["QLinear"](m=m, weight_bits=c["wb"], input_bits=c["ab"], symmetric=c["sym"], pretrained_initialized=True)
        x = T(det_normalish((c["B"], c["N"], c["I"]), sd + 2, 1.0))
        if not c["sym"]:
            x = x.abs()
        return q, x, None
    if k == "qmlp":
        m = ns["Mlp"](in_features=c["C"], hidden_features=c["Hd"], act_layer=nn.GELU)
        _fill(m.fc1.weight, sd, std=0.08)
        _fill(m.fc1.bias, sd + 1, lo=-0.1, hi=0.1)
        _fill(m.fc2.weight, sd + 2, std=0.04)
        _fill(m.fc2.bias, sd + 3, lo=-0.1, hi=0.1)
        q = ns["QMLP"](m=m, weight_bits=c["wb"], input_bits=c["ab"], act_layer=nn.GELU, pretrained_initialized=True)
        return q, T(det_normalish((c["B"], c["N"], c["C"]), sd + 4, 1.0)), None
    if k in ("qkr", "plain"):
        m = ns["Attention"](dim=c["C"], num_heads=c["H"], qkv_bias=True)
        _fill(m.qkv.weight, sd, std=0.06)
        _fill(m.qkv.bias, sd + 1, lo=-0.1, hi=0.1)
        _fill(m.proj.weight, sd + 2, std=0.05)
        _fill(m.proj.bias, sd + 3, lo=-0.1, hi=0.1)
        cls = ns["QAttention_qkreparam"] if k == "qkr" else ns["QAttention"]
        q = cls(m=m, weight_bits=c["wb"], input_bits=c["ab"], pretrained_initialized=True)
        return q, T(det_normalish((c["B"], c["N"], c["C"]), sd + 4, 1.0)), 0
    if k == "swin_qkr":
        m = ns["ShiftedWindowAttention"](c["C"], [7, 7], [c["shift"], c["shift"]], c["H"])
        _fill(m.qkv.weight, sd, std=0.08)
        _fill(m.qkv.bias, sd + 1, lo=-0.1, hi=0.1)
        _fill(m.proj.weight, sd + 2, std=0.06)
        _fill(m.proj.bias, sd + 3, lo=-0.1, hi=0.1)
        q = ns["QAttention_swin_qkreparam"](m=m, weight_bits=c["wb"], input_bits=c["ab"], pretrained_initialized=True)
        _fill(q.relative_position_bias_table, sd + 5, std=0.5)
        return q, T(det_normalish((c["B"], c["Hh"], c["Ww"], c["C"]), sd + 4, 1.0)), 0
    if k == "swin_mlp":
        m = ns["swin_MLP"](c["C"], [c["Hd"], c["C"]], activation_layer=nn.GELU, dropout=0.0)
        _fill(m[0].weight, sd, std=0.1)
        _fill(m[0].bias, sd + 1, lo=-0.1, hi=0.1)
        _fill(m[3].weight, sd + 2, std=0.05)
        _fill(m[3].bias, sd + 3, lo=-0.1, hi=0.1)
        q = ns["QMLP_swin"](m=m, weight_bits=c["wb"], input_bits=c["ab"], act_layer=nn.GELU, pretrained_initialized=True)
        return q, T(det_normalish((c["B"], c["Hh"], c["Ww"], c["C"]), sd + 4, 1.0)), None
    raise KeyError(k)


def upstream_grad(name, shape):
    return T(det_uniform(tuple(shape), CASES[name]["seed"] + 9, -1.0, 1.0))


def is_big_weight(t):
    return t.dim() >= 2 and t.numel() > 20_000


def randomize_small_params(mod, seed):
    """make_golden.randomize_offsets_and_scales: offsets non-zero, steps jittered (after the lazy init)."""
    k = 0
    for n, p in mod.named_parameters():
        k += 1
        if n.endswith("move_b4.bias") or n.endswith("move_aft.bias") or "move_" in n:
            p.data.copy_(T(det_uniform(tuple(p.shape), seed + k, -0.05, 0.05)))
        elif n.endswith(".s") or n == "s":
            p.data.mul_(T(det_uniform(tuple(p.shape), seed + k, 0.8, 1.25)))


def compact(d, key, arr):
    """Store `arr` under `key`: whole when small, else a strided sample + fp64 sum / l2 norm / max-abs."""
    a = np.ascontiguousarray(arr)
    if a.size <= FULL_MAX:
        d[key] = a
        return
    f = a.reshape(-1)
    d[key + "@sub"] = f[::STRIDE].copy()
    d[key + "@stat"] = np.array([f.astype(np.float64).sum(), np.sqrt((f.astype(np.float64) ** 2).sum()),
                                 np.abs(f).max(), a.size], dtype=np.float64)


def has(g, key):
    return key in g or key + "@sub" in g


def compare(actual, g, key):
    """Errors of `actual` (tensor) against the stored reference: dict with 'max' (max-abs error over max-abs reference, on
    the stored elements), 'l2' (relative l2 on the stored elements), 'bad' (share of the non-tiny stored elements whose own
    relative error exceeds 1e-3) and, for compact entries, 'norm' (relative difference of the full tensor's l2 norm)."""
    a = actual.detach().double().cpu().reshape(-1)
    out = {}
    if key in g:
        r = torch.from_numpy(np.ascontiguousarray(g[key])).double().reshape(-1)
    else:
        r = torch.from_numpy(g[key + "@sub"]).double()
        st = g[key + "@stat"]
        assert a.numel() == int(st[3]), (key, a.numel(), int(st[3]))
        out["norm"] = abs(float(a.norm()) - st[1]) / (st[1] + 1e-30)
        a = a[::STRIDE]
    assert a.numel() == r.numel(), (key, a.numel(), r.numel())
    den = float(r.abs().max()) + 1e-30
    out["max"] = float((a - r).abs().max()) / den
    out["l2"] = float((a - r).norm() / (r.norm() + 1e-30))
    big = r.abs() > 0.1 * den
    out["bad"] = float((((a - r).abs() / r.abs())[big] > 1e-3).double().mean()) if bool(big.any()) else 0.0
    return out
