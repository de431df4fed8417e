"""Generate golden vectors by running the REFERENCE ITSELF (nbasyl/OFQ at /root/reference) on CPU.

Run only in the build container (where /root/reference exists):
    python tests/golden/make_golden.py
Writes tests/golden/*.npz.  Inputs that are large come from detgen (seeded, platform-independent) and
are NOT stored; parameters (the reference module's state_dict after its lazy LSQ init), outputs and
autograd gradients are stored.  The reference publishes no tests/goldens of its own, so
these files are what pins oracle/ofq_oracle.py and, through it, the HIP path.
"""
import os
import sys
import copy
from functools import partial

import numpy as np
import torch
import torch.nn as nn

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from detgen import det_uniform, det_normalish, det_int  # noqa: E402
import ref_harness  # noqa: E402

torch.set_num_threads(4)
torch.manual_seed(0)

src = ref_harness.import_reference()
from src.quantization.quantizer.statsq import StatsQuantizer, StatsQuantizer_specific_4_qkreparam_cga  # noqa: E402
from src.quantization.quantizer import lsq as rlsq  # noqa: E402
from src.quantization.modules.qlinear import QLinear, QMLP, LSQ_QConv2d, LSQ_QLinear4head  # noqa: E402
from src.quantization.modules.attention import QAttention, QAttention_qkreparam, QAttention_qkreparam_4_cga  # noqa: E402
from src.quantization.modules.utils import replace_module_by_qmodule_deit  # noqa: E402
from src.quantization.utils import KDLossSoftandHard  # noqa: E402
from src.deit_vision_transformer import Attention, Mlp  # noqa: E402
from src.deit import DistilledVisionTransformer  # noqa: E402


def T(a):
    return torch.from_numpy(np.ascontiguousarray(a))


def npy(t):
    return t.detach().cpu().numpy().copy()


OUT_DIR = os.environ.get("OFQ_GOLDEN_OUT", HERE)       # tests/test_oracle_golden.py regenerates into a temp dir


def save(name, d):
    path = os.path.join(OUT_DIR, name + ".npz")
    np.savez_compressed(path, **d)
    print("wrote %s  (%d arrays, %.1f KB)" % (path, len(d), os.path.getsize(path) / 1024))


def randomize_offsets_and_scales(mod, seed):
    """After the lazy init: give every LearnableBias a non-zero value and jitter every LSQ `s`, so
    goldens exercise offsets and non-initial scales."""
    k = 0
    for n, p in mod.named_parameters():
        k += 1
        if n.endswith("move_b4.bias") or n.endswith("move_aft.bias") or "move_" in n:
            p.data.copy_(T(det_uniform(tuple(p.shape), seed + k, -0.05, 0.05)))
        elif n.endswith(".s") or n == "s":
            p.data.mul_(T(det_uniform(tuple(p.shape), seed + k, 0.8, 1.25)))


def run_module(mod, x, seed, out_index=None):
    """First forward = lazy LSQ init (like setup_alpha, train.py:997); then randomise offsets/scales,
    then a training forward+backward with a deterministic upstream gradient."""
    mod.train()
    with torch.no_grad():
        mod(x)
    randomize_offsets_and_scales(mod, seed)
    x = x.clone().requires_grad_(True)
    y = mod(x)
    if out_index is not None:
        y = y[out_index]
    g = T(det_uniform(tuple(y.shape), seed + 1000, -1.0, 1.0))
    (y * g).sum().backward()
    d = {"y": npy(y), "g": npy(g), "dx": npy(x.grad)}
    for n, p in mod.state_dict().items():
        d["p:" + n] = npy(p)
    for n, p in mod.named_parameters():
        if p.grad is not None:
            d["grad:" + n] = npy(p.grad)
    return d


# ------------------------------------------------------------------------------------------------
# G1 StatsQ (+ G10: the cga twin is bit-identical)
# ------------------------------------------------------------------------------------------------
def g1_statsq():
    torch.manual_seed(0)          # every generator starts from its own seed: any subset, any order, same files
    out = {}
    case = 0
    for (r, c) in [(8, 16), (12, 32), (48, 384), (6, 1536)]:
        for bits in (2, 3, 4):
            W = T(det_normalish((r, c), 11 + case, 0.02))
            # plant exact ties / clamp-edge values in row 0
            with torch.no_grad():
                s0 = 2 * W[0].abs().mean()
                W[0, 0] = 0.0
                W[0, 1] = float(s0) * 5.0
                W[0, 2] = -float(s0) * 5.0
            W.requires_grad_(True)
            q = StatsQuantizer(num_bits=bits, clip_learnable=False)
            y = q(W)
            g = T(det_uniform((r, c), 500 + case))
            (y * g).sum().backward()
            q2 = StatsQuantizer_specific_4_qkreparam_cga(num_bits=bits, clip_learnable=False)
            q2.train()
            W2 = W.detach().clone().requires_grad_(True)
            y2 = q2(W2)
            (y2 * g).sum().backward()
            assert torch.equal(y, y2) and torch.equal(W.grad, W2.grad), "G10: cga twin differs"
            n = float(2 ** (bits - 1))
            s = (2 * W.detach().abs().mean(dim=1, keepdim=True))
            L = torch.round(torch.clamp(W.detach() / s, -1.0, 1.0 - 1e-6) * n - 0.5)
            pre = "c%d:" % case
            out[pre + "shape"] = np.array([r, c, bits, 11 + case])
            out[pre + "W"] = npy(W)
            out[pre + "y"] = npy(y)
            out[pre + "s"] = npy(q.s)
            out[pre + "L"] = npy(L).astype(np.int8)
            out[pre + "g"] = npy(g)
            out[pre + "dW"] = npy(W.grad)
            case += 1
    out["ncases"] = np.array(case)
    save("g1_statsq", out)


# ------------------------------------------------------------------------------------------------
# G2 LSQ variants
# ------------------------------------------------------------------------------------------------
def _lsq_case(q, x, seed, plant=None):
    with torch.no_grad():
        q(x)                                           # lazy init
    s_init = npy(q.s)
    q.s.data.mul_(T(det_uniform(tuple(q.s.shape), seed, 0.8, 1.25)))
    if plant is not None:
        plant(x, q)
    x = x.clone().requires_grad_(True)
    y = q(x)
    g = T(det_uniform(tuple(y.shape), seed + 1, -1.0, 1.0))
    (y * g).sum().backward()
    return {"x": npy(x), "s_init": s_init, "s": npy(q.s), "y": npy(y), "g": npy(g), "dx": npy(x.grad),
            "ds": npy(q.s.grad), "lohi": np.array([q.thd_neg, q.thd_pos])}


def g2_lsq():
    torch.manual_seed(0)          # every generator starts from its own seed: any subset, any order, same files
    out = {}

    def put(name, d):
        for k, v in d.items():
            out[name + ":" + k] = v

    def plant_token(x, q):
        # exact .5 ties, exact lo/hi, and one scale below the 1e-5 floor
        with torch.no_grad():
            a = q.s.data
            x.view(-1, x.shape[-2], x.shape[-1])[0, 0, 0] = 0.5 * a[0]
            x.view(-1, x.shape[-2], x.shape[-1])[0, 0, 1] = 1.5 * a[0]
            x.view(-1, x.shape[-2], x.shape[-1])[0, 0, 2] = float(q.thd_pos) * a[0]
            x.view(-1, x.shape[-2], x.shape[-1])[0, 0, 3] = float(q.thd_neg) * a[0]
            a[1] = 3e-6

    for bits in (2, 3, 4, 8):
        for unsigned in (False, True):
            tag = "b%d%s" % (bits, "u" if unsigned else "s")
            x3 = T(det_normalish((3, 7, 16), 100 + bits, 1.0))
            if unsigned:
                x3 = x3.abs()
            put("token3d_" + tag, _lsq_case(rlsq.LsqQuantizer(bit=bits, all_positive=unsigned), x3, 200 + bits,
                                            plant_token))
            x4 = T(det_normalish((2, 3, 5, 8), 110 + bits, 1.0))
            if unsigned:
                x4 = x4.abs() * 0.3
            put("token4d_" + tag, _lsq_case(rlsq.LsqQuantizer(bit=bits, all_positive=unsigned), x4, 210 + bits))
        xv = T(det_normalish((3, 7, 16), 120 + bits, 1.0))
        put("chan3d_b%d" % bits, _lsq_case(rlsq.LsqQuantizer4v(bit=bits), xv, 220 + bits))
    # 8-bit specials
    ximg = T(det_normalish((2, 3, 12, 12), 130, 1.0))
    qi = rlsq.LsqQuantizer4img(bit=8)
    d = _lsq_case(qi, ximg, 230)
    d["signed"] = npy(qi.signed)
    put("img_signed", d)
    ximg = T(det_uniform((2, 3, 12, 12), 131, 0.0, 1.0))
    qi = rlsq.LsqQuantizer4img(bit=8)
    d = _lsq_case(qi, ximg, 231)
    d["signed"] = npy(qi.signed)
    put("img_unsigned", d)
    put("convw", _lsq_case(rlsq.LsqQuantizer4Conv2d(bit=8), T(det_normalish((6, 3, 4, 4), 132, 0.05)), 232))
    put("roww", _lsq_case(rlsq.LsqQuantizerWeight(bit=8), T(det_normalish((10, 24), 133, 0.05)), 233))
    put("tensor", _lsq_case(rlsq.LsqQuantizer4head_input(bit=8), T(det_normalish((4, 24), 134, 1.0)), 234))
    save("g2_lsq", out)


# ------------------------------------------------------------------------------------------------
# G3 / G5 QLinear, QMLP
# ------------------------------------------------------------------------------------------------
def _linear(i, o, seed):
    m = nn.Linear(i, o)
    with torch.no_grad():
        m.weight.copy_(T(det_normalish((o, i), seed, 0.05)))
        m.bias.copy_(T(det_uniform((o,), seed + 1, -0.1, 0.1)))
    return m


def g3_qlinear():
    torch.manual_seed(0)          # every generator starts from its own seed: any subset, any order, same files
    out = {}
    cases = [("toy_w2a2", 2, 5, 16, 24, 2, 2, True), ("toy_w4a4", 2, 5, 16, 24, 4, 4, True),
             ("toy_w3a3_unsigned", 2, 5, 16, 24, 3, 3, False), ("n198_w2a2", 2, 198, 96, 160, 2, 2, True)]
    for k, (name, B, N, I, O, wb, ab, sym) in enumerate(cases):
        m = _linear(I, O, 300 + 10 * k)
        q = QLinear(m=m, weight_bits=wb, input_bits=ab, symmetric=sym, pretrained_initialized=True)
        x = T(det_normalish((B, N, I), 301 + 10 * k, 1.0))
        if not sym:
            x = x.abs()
        d = run_module(q, x, 302 + 10 * k)
        d["meta"] = np.array([B, N, I, O, wb, ab, int(sym), 301 + 10 * k])
        for kk, v in d.items():
            out[name + ":" + kk] = v
    save("g3_qlinear", out)


def g5_qmlp():
    torch.manual_seed(0)          # every generator starts from its own seed: any subset, any order, same files
    out = {}
    for k, (name, B, N, C, Hd, wb, ab) in enumerate([("toy_w2a2", 2, 5, 16, 64, 2, 2), ("toy_w4a4", 2, 7, 16, 48, 4, 4)]):
        m = Mlp(in_features=C, hidden_features=Hd, act_layer=nn.GELU)
        with torch.no_grad():
            m.fc1.weight.copy_(T(det_normalish((Hd, C), 400 + 10 * k, 0.2)))
            m.fc1.bias.copy_(T(det_uniform((Hd,), 401 + 10 * k, -0.1, 0.1)))
            m.fc2.weight.copy_(T(det_normalish((C, Hd), 402 + 10 * k, 0.1)))
            m.fc2.bias.copy_(T(det_uniform((C,), 403 + 10 * k, -0.1, 0.1)))
        q = QMLP(m=m, weight_bits=wb, input_bits=ab, act_layer=nn.GELU, pretrained_initialized=True)
        x = T(det_normalish((B, N, C), 404 + 10 * k, 1.0))
        d = run_module(q, x, 405 + 10 * k)
        d["meta"] = np.array([B, N, C, Hd, wb, ab, 404 + 10 * k])
        for kk, v in d.items():
            out[name + ":" + kk] = v
    save("g5_qmlp", out)


# ------------------------------------------------------------------------------------------------
# G4 attention (plain, QKR, QKR-cga twin)
# ------------------------------------------------------------------------------------------------
def _attention(C, H, seed):
    m = Attention(dim=C, num_heads=H, qkv_bias=True)
    with torch.no_grad():
        m.qkv.weight.copy_(T(det_normalish((3 * C, C), seed, 0.15)))
        m.qkv.bias.copy_(T(det_uniform((3 * C,), seed + 1, -0.1, 0.1)))
        m.proj.weight.copy_(T(det_normalish((C, C), seed + 2, 0.1)))
        m.proj.bias.copy_(T(det_uniform((C,), seed + 3, -0.1, 0.1)))
    return m


def g4_attention():
    torch.manual_seed(0)          # every generator starts from its own seed: any subset, any order, same files
    out = {}
    shapes = [("toy", 2, 7, 32, 2), ("n198", 2, 198, 64, 2)]
    k = 0
    for (sname, B, N, C, H) in shapes:
        for (wb, ab) in [(2, 2), (4, 4)]:
            for kind, cls in [("plain", QAttention), ("qkr", QAttention_qkreparam), ("qkrcga", QAttention_qkreparam_4_cga)]:
                if kind == "qkrcga" and (sname != "toy" or wb != 2):
                    continue
                m = _attention(C, H, 500 + 10 * k)
                q = cls(m=m, weight_bits=wb, input_bits=ab, pretrained_initialized=True)
                x = T(det_normalish((B, N, C), 505 + 10 * k, 1.0))
                d = run_module(q, x, 506 + 10 * k, out_index=0)
                d["meta"] = np.array([B, N, C, H, wb, ab, 505 + 10 * k])
                name = "%s_%s_w%da%d" % (kind, sname, wb, ab)
                if sname == "n198":
                    # keep the fixture small: parameters + outputs + gradients only, the input is regenerable
                    pass
                for kk, v in d.items():
                    out[name + ":" + kk] = v
                k += 1
    save("g4_attention", out)


# ------------------------------------------------------------------------------------------------
# G6 patch embed + head   (LearnableBias4img is hard-wired to 224x224, qlinear.py:163-164)
# ------------------------------------------------------------------------------------------------
def g6_stem_head():
    torch.manual_seed(0)          # every generator starts from its own seed: any subset, any order, same files
    out = {}
    conv = nn.Conv2d(3, 8, kernel_size=16, stride=16)
    with torch.no_grad():
        conv.weight.copy_(T(det_normalish((8, 3, 16, 16), 600, 0.05)))
        conv.bias.copy_(T(det_uniform((8,), 601, -0.1, 0.1)))
    for name, lo in [("signed", -2.0), ("unsigned", 0.0)]:
        q = LSQ_QConv2d(m=copy.deepcopy(conv), weight_bits=8, input_bits=8, weight_quant_method="lsq",
                        input_quant_method="lsq", pretrained_initialized=True)
        img = T(det_uniform((1, 3, 224, 224), 602, lo, 2.0))
        d = run_module(q, img, 603)
        d["meta"] = np.array([1, 8, 602, int(lo < 0)])
        d["img_lohi"] = np.array([lo, 2.0], dtype=np.float32)
        for kk, v in d.items():
            out["conv_" + name + ":" + kk] = v
    m = _linear(24, 10, 610)
    q = LSQ_QLinear4head(m=m, weight_bits=8, input_bits=8, weight_quant_method="lsq", input_quant_method="lsq",
                         pretrained_initialized=True)
    x = T(det_normalish((4, 24), 611, 1.0))
    d = run_module(q, x, 612)
    d["meta"] = np.array([4, 24, 10, 611])
    for kk, v in d.items():
        out["head:" + kk] = v
    save("g6_stem_head", out)


# ------------------------------------------------------------------------------------------------
# G7 tiny distilled DeiT, full training step loss + all grads (plain and QKR)
# ------------------------------------------------------------------------------------------------
def _qconfigs(names, wb, ab):
    qc = {}
    for n in names:
        qc[n] = {"weight": {"mode": "statsq", "bit": wb, "all_positive": False, "symmetric": True, "per_channel": True,
                            "normalize_first": False, "learnable": True},
                 "act": {"enable": True, "mode": "lsq", "bit": ab, "per_channel": True, "normalize_first": False,
                         "learnable": True},
                 "q_attn_dropout": False, "act_layer": nn.GELU}
    return qc


def g7_tiny_deit():
    torch.manual_seed(0)          # every generator starts from its own seed: any subset, any order, same files
    out = {}
    for k, (name, qkr, wb, ab) in enumerate([("plain_w4a4", False, 4, 4), ("qkr_w2a2", True, 2, 2)]):
        depth, dim, heads = 2, 32, 2
        torch.manual_seed(1234 + k)
        model = DistilledVisionTransformer(img_size=224, patch_size=16, embed_dim=dim, depth=depth, num_heads=heads,
                                           mlp_ratio=2, qkv_bias=True, num_classes=10,
                                           norm_layer=partial(nn.LayerNorm, eps=1e-6), act_layer=nn.GELU)
        # deterministic, non-degenerate weights
        with torch.no_grad():
            for i, (n, p) in enumerate(model.named_parameters()):
                if p.dim() >= 2 and "norm" not in n:
                    p.copy_(T(det_normalish(tuple(p.shape), 700 + 50 * k + i, 0.08)))
                elif "norm" in n and n.endswith("weight"):
                    p.copy_(T(det_uniform(tuple(p.shape), 700 + 50 * k + i, 0.8, 1.2)))
                else:
                    p.copy_(T(det_uniform(tuple(p.shape), 700 + 50 * k + i, -0.1, 0.1)))
        names = ["patch_embed.proj"] + sum([["blocks.%d.attn" % i, "blocks.%d.mlp" % i] for i in range(depth)], []) + \
                ["head", "head_dist"]
        model = replace_module_by_qmodule_deit(model, _qconfigs(names, wb, ab), pretrained_initialized=True,
                                               qk_reparam=qkr, qk_reparam_type=0)
        B = 2
        img = T(det_uniform((B, 3, 224, 224), 760 + k, -2.0, 2.0))
        target = T(det_int((B,), 761 + k, 10))
        soft = T(det_normalish((B, 10), 762 + k, 2.0))
        model.eval()
        with torch.no_grad():
            model(img)                                    # setup_alpha (train.py:997-1010)
        randomize_offsets_and_scales(model, 770 + k)
        model.train()
        (cls_o, dist_o), _ = model(img)
        loss = KDLossSoftandHard()((cls_o, dist_o), target, soft)
        loss.backward()
        d = {"cls": npy(cls_o), "dist": npy(dist_o), "loss": npy(loss), "target": npy(target), "soft": npy(soft),
             "meta": np.array([B, depth, dim, heads, wb, ab, int(qkr), 760 + k, 10, 2])}
        for n, p in model.state_dict().items():
            d["p:" + n] = npy(p)
        for n, p in model.named_parameters():
            if p.grad is not None:
                d["grad:" + n] = npy(p.grad)
        model.eval()
        with torch.no_grad():
            ev, _ = model(img)
        d["eval_logits"] = npy(ev)
        for kk, v in d.items():
            out[name + ":" + kk] = v
    save("g7_tiny_deit", out)


# ------------------------------------------------------------------------------------------------
# G8 CGA: freeze idx, masked grad, restored weights after one AdamW step
# ------------------------------------------------------------------------------------------------
def g8_cga():
    torch.manual_seed(0)          # every generator starts from its own seed: any subset, any order, same files
    import importlib.util
    # cga.py is a script that imports timm training utilities at module import; the function we need
    # (cga.py:450-469) only uses torch/numpy, so execute just that function's source text.
    import ast
    srcp = os.path.join(ref_harness.REFERENCE_ROOT, "cga.py")
    tree = ast.parse(open(srcp).read())
    fn = [n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name == "freeze_outside_boundary_weight_idx"][0]
    ns = {"torch": torch, "np": np}
    exec(compile(ast.Module(body=[fn], type_ignores=[]), srcp, "exec"), ns)
    freeze_fn = ns["freeze_outside_boundary_weight_idx"]
    out = {}
    case = 0
    for (r, c, bits, br) in [(8, 64, 2, 0.005), (16, 96, 2, 0.05), (8, 64, 3, 0.02), (8, 64, 4, 0.05)]:
        W = T(det_normalish((r, c), 800 + case, 0.02))
        with torch.no_grad():
            # plant weights right at / around the rounding boundaries of row 0
            s0 = float(2 * W[0].abs().mean())
            n = float(2 ** (bits - 1))
            for j, off in enumerate([0.0, br * 0.5, -br * 0.5, br * 1.5, -br * 1.5]):
                W[0, j] = s0 * ((0.0 + 0.5 + off + 0.5) / n)      # b4_round = 0.5 + off  (boundary between 0 and 1)
        frz = freeze_fn(W, bits, boundaryRange=br)
        g = T(det_uniform((r, c), 810 + case))
        gm = g * frz * 0.0 + g * (1 - frz)                        # cga.py:962
        Wp = torch.nn.Parameter(W.clone())
        opt = torch.optim.AdamW([Wp], lr=1e-3, weight_decay=0.05)
        Wp.grad = gm.clone()
        saved = (Wp * frz).detach().clone()                       # cga.py:964
        opt.step()
        with torch.no_grad():
            new_w = Wp.detach().clone() * (1 - frz) + saved       # cga.py:994-997
        pre = "c%d:" % case
        out[pre + "meta"] = np.array([r, c, bits, 800 + case])
        out[pre + "br"] = np.array(br, dtype=np.float64)
        out[pre + "W"] = npy(W)
        out[pre + "frz"] = npy(frz)
        out[pre + "g"] = npy(g)
        out[pre + "gm"] = npy(gm)
        out[pre + "W_after"] = npy(new_w)
        case += 1
    out["ncases"] = np.array(case)
    save("g8_cga", out)


# ------------------------------------------------------------------------------------------------
# G9 Swin: window attention (plain / QKR / QKR-cga, shifted and not), 4-D QLinear (`reduction`), QMLP_swin,
#          and a tiny full Swin model step
# ------------------------------------------------------------------------------------------------
def g9_swin():
    torch.manual_seed(0)          # every generator starts from its own seed: any subset, any order, same files
    from src.swin import ShiftedWindowAttention, SwinTransformer
    from src.quantization.modules.swin_attention_and_mlp import (QAttention_swin, QAttention_swin_qkreparam,
                                                                 QAttention_swin_qkreparam_4_cga, QMLP_swin)
    from src.quantization.modules.utils import replace_module_by_qmodule_swin
    from torchvision.ops.misc import MLP as swin_MLP
    out = {}
    k = 0
    for (kind, cls) in [("plain", QAttention_swin), ("qkr", QAttention_swin_qkreparam), ("qkrcga", QAttention_swin_qkreparam_4_cga)]:
        for shift in (0, 3):
            for (wb, ab) in [(2, 2), (4, 4)]:
                if kind == "qkrcga" and (shift == 0 or wb != 2):
                    continue
                B, Hh, Ww, C, H = 2, 14, 13, 24, 3           # W = 13 exercises the right-padding path
                m = ShiftedWindowAttention(C, [7, 7], [shift, shift], H)
                with torch.no_grad():
                    m.qkv.weight.copy_(T(det_normalish((3 * C, C), 900 + 10 * k, 0.2)))
                    m.qkv.bias.copy_(T(det_uniform((3 * C,), 901 + 10 * k, -0.1, 0.1)))
                    m.proj.weight.copy_(T(det_normalish((C, C), 902 + 10 * k, 0.15)))
                    m.proj.bias.copy_(T(det_uniform((C,), 903 + 10 * k, -0.1, 0.1)))
                q = cls(m=m, weight_bits=wb, input_bits=ab, pretrained_initialized=True)
                with torch.no_grad():
                    q.relative_position_bias_table.copy_(T(det_normalish(tuple(q.relative_position_bias_table.shape),
                                                                         904 + 10 * k, 0.5)))
                x = T(det_normalish((B, Hh, Ww, C), 905 + 10 * k, 1.0))
                d = run_module(q, x, 906 + 10 * k, out_index=0)
                d["meta"] = np.array([B, Hh, Ww, C, H, wb, ab, shift, 905 + 10 * k])
                name = "attn_%s_s%d_w%da%d" % (kind, shift, wb, ab)
                for kk, v in d.items():
                    out[name + ":" + kk] = v
                k += 1
    # config C4's bit-width (Swin-T W3A3, BASELINE.json configs[3]): shifted windows, all three attention variants
    for j, (kind, cls) in enumerate([("plain", QAttention_swin), ("qkr", QAttention_swin_qkreparam),
                                     ("qkrcga", QAttention_swin_qkreparam_4_cga)]):
        B, Hh, Ww, C, H, shift, wb, ab = 2, 14, 13, 24, 3, 3, 3, 3
        sd = 2000 + 10 * j
        m = ShiftedWindowAttention(C, [7, 7], [shift, shift], H)
        with torch.no_grad():
            m.qkv.weight.copy_(T(det_normalish((3 * C, C), sd, 0.2)))
            m.qkv.bias.copy_(T(det_uniform((3 * C,), sd + 1, -0.1, 0.1)))
            m.proj.weight.copy_(T(det_normalish((C, C), sd + 2, 0.15)))
            m.proj.bias.copy_(T(det_uniform((C,), sd + 3, -0.1, 0.1)))
        q = cls(m=m, weight_bits=wb, input_bits=ab, pretrained_initialized=True)
        with torch.no_grad():
            q.relative_position_bias_table.copy_(T(det_normalish(tuple(q.relative_position_bias_table.shape), sd + 4, 0.5)))
        x = T(det_normalish((B, Hh, Ww, C), sd + 5, 1.0))
        d = run_module(q, x, sd + 6, out_index=0)
        d["meta"] = np.array([B, Hh, Ww, C, H, wb, ab, shift, sd + 5])
        for kk, v in d.items():
            out["attn_%s_s%d_w%da%d:" % (kind, shift, wb, ab) + kk] = v
    # QLinear on a 4-D input (PatchMerging.reduction: bias-less source -> default-initialised QLinear bias)
    lin = nn.Linear(48, 24, bias=False)
    with torch.no_grad():
        lin.weight.copy_(T(det_normalish((24, 48), 990, 0.1)))
    q = QLinear(m=lin, weight_bits=3, input_bits=3, pretrained_initialized=True)
    with torch.no_grad():
        q.bias.copy_(T(det_uniform((24,), 991, -0.1, 0.1)))
    d = run_module(q, T(det_normalish((2, 5, 6, 48), 992, 1.0)), 993)
    d["meta"] = np.array([2, 5, 6, 48, 24, 3, 3, 992])
    for kk, v in d.items():
        out["reduction4d:" + kk] = v
    mm = swin_MLP(24, [48, 24], activation_layer=nn.GELU, inplace=None, dropout=0.0)
    with torch.no_grad():
        mm[0].weight.copy_(T(det_normalish((48, 24), 994, 0.2)))
        mm[0].bias.copy_(T(det_uniform((48,), 995, -0.1, 0.1)))
        mm[3].weight.copy_(T(det_normalish((24, 48), 996, 0.15)))
        mm[3].bias.copy_(T(det_uniform((24,), 997, -0.1, 0.1)))
    q = QMLP_swin(m=mm, weight_bits=2, input_bits=2, act_layer=nn.GELU, pretrained_initialized=True)
    d = run_module(q, T(det_normalish((2, 5, 6, 24), 998, 1.0)), 999)
    d["meta"] = np.array([2, 5, 6, 24, 48, 2, 2, 998])
    for kk, v in d.items():
        out["mlp4d:" + kk] = v
    save("g9_swin_modules", out)

    # tiny full Swin model
    torch.manual_seed(1)          # independent of how many module cases ran above
    out = {}
    for k, (name, qkr, wb, ab) in enumerate([("plain_w4a4", False, 4, 4), ("qkr_w2a2", True, 2, 2), ("qkr_w3a3", True, 3, 3)]):
        depths, heads, dim, ncls = [2, 2], [2, 4], 16, 10
        model = SwinTransformer(patch_size=[4, 4], embed_dim=dim, depths=depths, num_heads=heads, window_size=[7, 7],
                                num_classes=ncls)
        with torch.no_grad():
            for i, (n, p) in enumerate(model.named_parameters()):
                if p.dim() >= 2 and "norm" not in n:
                    p.copy_(T(det_normalish(tuple(p.shape), 1100 + 80 * k + i, 0.15)))
                elif "norm" in n and n.endswith("weight"):
                    p.copy_(T(det_uniform(tuple(p.shape), 1100 + 80 * k + i, 0.8, 1.2)))
                else:
                    p.copy_(T(det_uniform(tuple(p.shape), 1100 + 80 * k + i, -0.1, 0.1)))
        names = ["features.0.0"]
        fi = 1
        for si, dd in enumerate(depths):
            for li in range(dd):
                names += ["features.%d.%d.attn" % (fi, li), "features.%d.%d.mlp" % (fi, li)]
            fi += 1
            if si < len(depths) - 1:
                names.append("features.%d.reduction" % fi)
                fi += 1
        names.append("head")
        model = replace_module_by_qmodule_swin(model, _qconfigs(names, wb, ab), pretrained_initialized=True,
                                               qk_reparam=qkr, qk_reparam_type=0)
        B = 1
        img = T(det_uniform((B, 3, 224, 224), 1260 + k, -2.0, 2.0))
        target = T(det_int((B,), 1261 + k, ncls))
        soft = T(det_normalish((B, ncls), 1262 + k, 2.0))
        model.eval()
        with torch.no_grad():
            model(img)
        randomize_offsets_and_scales(model, 1270 + k)
        with torch.no_grad():
            for n, p in model.named_parameters():
                if n.endswith("relative_position_bias_table"):
                    p.copy_(T(det_normalish(tuple(p.shape), 1280 + k, 0.5)))
        model.train()
        logits, _ = model(img)
        loss = KDLossSoftandHard()(logits, target, soft)
        loss.backward()
        d = {"logits": npy(logits), "loss": npy(loss), "target": npy(target), "soft": npy(soft),
             "meta": np.array([B, dim, wb, ab, int(qkr), 1260 + k, ncls] + depths + heads)}
        for n, p in model.state_dict().items():
            d["p:" + n] = npy(p)
        for n, p in model.named_parameters():
            if p.grad is not None:
                d["grad:" + n] = npy(p.grad)
        for kk, v in d.items():
            out[name + ":" + kk] = v
    save("g9_swin_tiny", out)


# ------------------------------------------------------------------------------------------------
# G10 the modules at PRODUCTION dimensions (the shapes that select the production kernel instances); weights from
#     detgen seeds (not stored), outputs / gradients in compact form -- tests/golden/prodcases.py
# ------------------------------------------------------------------------------------------------
def g10_prod():
    torch.manual_seed(0)
    import prodcases as PC
    from src.swin import ShiftedWindowAttention
    from src.quantization.modules.swin_attention_and_mlp import QAttention_swin_qkreparam, QMLP_swin
    from torchvision.ops.misc import MLP as swin_MLP
    ns = {"QLinear": QLinear, "QMLP": QMLP, "Mlp": Mlp, "Attention": Attention, "QAttention": QAttention,
          "QAttention_qkreparam": QAttention_qkreparam, "ShiftedWindowAttention": ShiftedWindowAttention,
          "QAttention_swin_qkreparam": QAttention_swin_qkreparam, "QMLP_swin": QMLP_swin, "swin_MLP": swin_MLP}
    out = {}
    for name in PC.CASES:
        q, x, oi = PC.build(name, ns)
        q.train()
        with torch.no_grad():
            q(x)                                              # lazy LSQ init (setup_alpha, train.py:997)
        PC.randomize_small_params(q, PC.CASES[name]["seed"] + 100)
        xg = x.clone().requires_grad_(True)
        y = q(xg)
        if oi is not None:
            y = y[oi]
        g = PC.upstream_grad(name, y.shape)
        (y * g).sum().backward()
        # tie-free by construction: the same module in fp64, and in fp32 on one thread (another summation order), must give
        # the same outputs and gradients; otherwise some value sits on a rounding tie and the case has no single answer
        def rerun(mod, xin):
            xr = xin.clone().requires_grad_(True)
            yr = mod(xr)
            yr = yr[oi] if oi is not None else yr
            mod.zero_grad()
            (yr * g.to(yr.dtype)).sum().backward()
            r = {"y": yr.detach().double(), "dx": xr.grad.double()}
            r.update({n: p.grad.double().clone() for n, p in mod.named_parameters() if p.grad is not None and "move_" not in n})
            return r
        ref32 = {"y": y.detach().double(), "dx": xg.grad.double()}
        ref32.update({n: p.grad.double().clone() for n, p in q.named_parameters() if p.grad is not None and "move_" not in n})
        r64 = rerun(copy.deepcopy(q).double(), x.double())
        torch.set_num_threads(1)
        r1 = rerun(copy.deepcopy(q), x)
        torch.set_num_threads(4)
        for kk in ref32:
            e64 = float((ref32[kk] - r64[kk]).norm() / (r64[kk].norm() + 1e-30))
            e1 = float((ref32[kk] - r1[kk]).norm() / (ref32[kk].norm() + 1e-30))
            assert e64 < 1e-5 and e1 < 1e-5, "case %s is not tie-free (%s: fp64 %.1e, 1 thread %.1e): pick another seed" % (
                name, kk, e64, e1)
        d = {}
        PC.compact(d, "y", npy(y))
        PC.compact(d, "dx", npy(xg.grad))
        for n, p in q.state_dict().items():
            if PC.is_big_weight(p):
                d["w2:" + n] = np.array(float(p.double().norm()))      # regenerable from the seeds: only its norm
            else:
                d["p:" + n] = npy(p)
        for n, p in q.named_parameters():
            if p.grad is not None:
                PC.compact(d, "grad:" + n, npy(p.grad))
        for kk, v in d.items():
            out[name + ":" + kk] = v
        print("  %-14s y%s  %d entries" % (name, tuple(y.shape), len(d)))
    save("g10_prod", out)


# ------------------------------------------------------------------------------------------------
# G11 the fp32 KD teacher (train.py:428-442, :906-910): the UNQUANTISED distilled DeiT of the reference, train and
#     eval mode outputs.  Parameters from detgen seeds by parameter index (not stored).
# ------------------------------------------------------------------------------------------------
TEACHER_CASES = {"tiny_d12": dict(dim=192, depth=12, heads=3, B=2, ncls=1000, seed=4000),
                 "small_d2": dict(dim=384, depth=2, heads=6, B=2, ncls=1000, seed=4100)}


def teacher_fill(model, seed):
    with torch.no_grad():
        for i, (n, p) in enumerate(model.named_parameters()):
            if p.dim() >= 2 and "norm" not in n:
                p.copy_(T(det_normalish(tuple(p.shape), seed + i, 0.05)))
            elif "norm" in n and n.endswith("weight"):
                p.copy_(T(det_uniform(tuple(p.shape), seed + i, 0.8, 1.2)))
            else:
                p.copy_(T(det_uniform(tuple(p.shape), seed + i, -0.1, 0.1)))


def g11_teacher():
    torch.manual_seed(0)
    out = {}
    for name, c in TEACHER_CASES.items():
        model = DistilledVisionTransformer(img_size=224, patch_size=16, embed_dim=c["dim"], depth=c["depth"],
                                           num_heads=c["heads"], mlp_ratio=4, qkv_bias=True, num_classes=c["ncls"],
                                           norm_layer=partial(nn.LayerNorm, eps=1e-6), act_layer=nn.GELU)
        teacher_fill(model, c["seed"])
        img = T(det_uniform((c["B"], 3, 224, 224), c["seed"] + 900, -2.0, 2.0))
        model.train()                                        # the reference never puts its teacher in eval mode
        with torch.no_grad():
            (cls_o, dist_o), _ = model(img)
            model.eval()
            ev, _ = model(img)
        out[name + ":cls"] = npy(cls_o)
        out[name + ":dist"] = npy(dist_o)
        out[name + ":eval"] = npy(ev)
        out[name + ":meta"] = np.array([c["dim"], c["depth"], c["heads"], c["B"], c["ncls"], c["seed"]])
        out[name + ":w2"] = np.array(float(sum(p.double().pow(2).sum() for p in model.parameters()) ** 0.5))
    save("g11_teacher", out)


if __name__ == "__main__":
    which = sys.argv[1:] or ["g1", "g2", "g3", "g4", "g5", "g6", "g7", "g8", "g9", "g10", "g11"]
    fns = {"g1": g1_statsq, "g2": g2_lsq, "g3": g3_qlinear, "g4": g4_attention, "g5": g5_qmlp, "g6": g6_stem_head,
           "g7": g7_tiny_deit, "g8": g8_cga, "g9": g9_swin, "g10": g10_prod, "g11": g11_teacher}
    for w in which:
        fns[w]()
