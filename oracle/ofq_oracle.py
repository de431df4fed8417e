"""CPU oracle for the OFQ fake-quantised ViT hot path.  TEST INFRASTRUCTURE ONLY.

This is a *restatement* (not a copy) of the reference algorithm (nbasyl/OFQ, /root/reference) as flat
functions over explicit parameter dicts, written with eager torch-CPU ops in the reference's op order
so that values and autograd gradients are bit-comparable with the reference's own PyTorch-CPU path.
Every function cites the reference file:line it follows.

Who may import this file: tests/, __graft_entry__.smoke() and bench.py's `cpu_baseline` leg, and only
as the checker / the timed CPU baseline.  The product (ofq_amd/) never imports it and has no CPU
fallback: it raises when the HIP library or a HIP device tensor is missing.

Pinning: the reference has no tests or golden vectors of its own.  This oracle is
pinned against outputs of the reference itself, generated in the build container by importing
/root/reference (tests/golden/make_golden.py -> tests/golden/*.npz, checked by tests/test_oracle_golden.py).

Parameter dicts use the reference's state-dict key names, e.g. for a QLinear:
  weight, bias, input_quant_fn.s, move_b4.bias, move_aft.bias
"""
import math

import torch
import torch.nn.functional as F

# ----------------------------------------------------------------------------------------------
# small STE helpers (reference: quantizer/lsq.py:6-18)
# ----------------------------------------------------------------------------------------------


def _grad_scale(x, scale):
    # lsq.py:6-9 : value x, gradient scaled by `scale`
    yg = x * scale
    return (x - yg).detach() + yg


def _round_pass(x):
    # lsq.py:11-14 : value round(x) (RNE), gradient identity
    return (x.round() - x).detach() + x


def _clip_eps(x, eps=1e-5):
    # lsq.py:16-18 : value max(x, eps), gradient 1 everywhere
    x_clip = torch.where(x > eps, x, torch.tensor(eps, dtype=x.dtype))
    return x - x.detach() + x_clip.detach()


def lsq_bounds(bits, unsigned):
    # lsq.py:519-534
    if unsigned:
        return (0, 1) if bits == 1 else (0, 2 ** bits - 1)
    return (-1, 1) if bits == 1 else (-(2 ** (bits - 1)), 2 ** (bits - 1) - 1)


# ----------------------------------------------------------------------------------------------
# StatsQ weight quantiser (statsq.py:133-150)
# ----------------------------------------------------------------------------------------------


def statsq(W, bits, s=None):
    """Returns (W_hat with STE gradient, integer levels L, scale s[rows,1]).  statsq.py:138-148.
    s: test hook, see cga_freeze_idx."""
    assert W.dim() == 2
    if s is None:
        s = (2 * torch.mean(W.abs(), dim=1, keepdim=True)).detach()      # :138, :142
    else:
        s = s.detach().reshape(-1, 1).to(W.dtype)
    v = W / s                                                            # :144
    c = torch.clamp(v, min=-(2.0 / 2), max=(2.0 / 2) - 1e-6)             # :145  clip_val = 2.0 (:126-128)
    n = float(2 ** (bits - 1))                                           # :146
    L = torch.round(c * n - 0.5)                                         # :147
    Wq = s * ((L + 0.5) / n)                                             # :147
    out = Wq.detach() - W.detach() + W                                   # :148  STE: dW = g everywhere
    return out, L.detach().to(torch.int32), s


# ----------------------------------------------------------------------------------------------
# LSQ family.  One generic body (lsq.py:571-602 and the identical bodies at :72-101, :336-373,
# :419-437, :489-505, :757-792); the variants differ only in where `s` broadcasts and in M.
# ----------------------------------------------------------------------------------------------


def _lsq_core(x, alpha, lo, hi, gscale):
    s_scale = _grad_scale(_clip_eps(alpha), gscale)                      # lsq.py:593
    v = x / s_scale                                                      # :595
    u = torch.clamp(v, lo, hi)                                           # :599
    q = _round_pass(u)                                                   # :600
    return q * s_scale                                                   # :601


def lsq_token(x, s, bits, unsigned):
    """LsqQuantizer (lsq.py:515-610): s has length x.shape[-2], broadcast as s.unsqueeze(-1) (:575)."""
    lo, hi = lsq_bounds(bits, unsigned)
    if x.dim() == 3:
        M = x.shape[0] * x.shape[-1]                                     # :584
    elif x.dim() == 2:
        M = x.shape[-1]                                                  # :586
    else:
        M = x.shape[0] * x.shape[1] * x.shape[-1]                        # :588
    return _lsq_core(x, s.unsqueeze(-1), lo, hi, 1.0 / math.sqrt(hi * M))


def lsq_token_init(x, bits, unsigned):
    """init_from (lsq.py:544-551): nested means, k = 2 signed / 4 unsigned."""
    lo, hi = lsq_bounds(bits, unsigned)
    k = 4 if unsigned else 2
    a = x.detach().abs().mean(dim=-1)
    if x.dim() == 3:
        a = a.mean(dim=0)
    elif x.dim() == 4:
        a = a.mean(dim=0).mean(dim=0)
    return k * a / (hi ** 0.5)


def lsq_channel(x, s, bits, unsigned=False):
    """LsqQuantizer4v (lsq.py:701-800): s per last dim; M = product of leading dims (:775-778)."""
    lo, hi = lsq_bounds(bits, unsigned)
    M = x.numel() // x.shape[-1]
    return _lsq_core(x, s, lo, hi, 1.0 / math.sqrt(hi * M))


def lsq_channel_init(x, bits, unsigned=False):
    lo, hi = lsq_bounds(bits, unsigned)                                  # lsq.py:732-737
    k = 4 if unsigned else 2
    a = x.detach().abs()
    while a.dim() > 1:
        a = a.mean(dim=0)
    return k * a / (hi ** 0.5)


def lsq_img(x, s, signed, bits=8):
    """LsqQuantizer4img (lsq.py:306-382): s per input channel (dim 1) of (B,C,H,W); signedness latched."""
    lo, hi = lsq_bounds(bits, not signed)                                # :341-355
    M = x.shape[0] * x.shape[2] * x.shape[3]                             # :363
    return _lsq_core(x, s.view(1, -1, 1, 1), lo, hi, 1.0 / math.sqrt(hi * M))


def lsq_img_init(x, signed, bits=8):
    lo, hi = lsq_bounds(bits, not signed)
    k = 2 if signed else 2                                               # all_positive is False at the call site (qlinear.py:154) -> factor 2
    return k * x.detach().abs().mean(dim=-1).mean(dim=-1).mean(dim=0) / (hi ** 0.5)   # :322


def lsq_convw(W, s, bits=8):
    """LsqQuantizer4Conv2d (lsq.py:384-446): signed, s per out-channel of (O,I,kh,kw)."""
    lo, hi = lsq_bounds(bits, False)
    M = W.shape[1] * W.shape[2] * W.shape[3]                             # :427
    return _lsq_core(W, s.view(-1, 1, 1, 1), lo, hi, 1.0 / math.sqrt(hi * M))


def lsq_convw_init(W, bits=8):
    lo, hi = lsq_bounds(bits, False)
    return 2 * W.detach().abs().mean(dim=-1).mean(dim=-1).mean(dim=-1) / (hi ** 0.5)  # :405


def lsq_roww(W, s, bits=8):
    """LsqQuantizerWeight (lsq.py:20-109), per_channel: s per row of a 2-D weight; M = in_features (:87)."""
    lo, hi = lsq_bounds(bits, False)
    return _lsq_core(W, s.unsqueeze(-1), lo, hi, 1.0 / math.sqrt(hi * W.shape[-1]))


def lsq_roww_init(W, bits=8):
    lo, hi = lsq_bounds(bits, False)
    return 2 * W.detach().abs().mean(dim=-1) / (hi ** 0.5)               # :54


def lsq_tensor(x, s, bits=8):
    """LsqQuantizer4head_input (lsq.py:448-513): one scalar s; M = numel (:494)."""
    lo, hi = lsq_bounds(bits, False)
    return _lsq_core(x, s, lo, hi, 1.0 / math.sqrt(hi * x.numel()))


def lsq_tensor_init(x, bits=8):
    lo, hi = lsq_bounds(bits, False)
    return (x.detach().abs().mean() * 2 / (hi ** 0.5)).reshape(1)        # :480


def lsq_effective_scale(alpha, gscale):
    """The scale VALUE the reference divides by (lsq.py:593): clip() returns exactly max(alpha,1e-5), but
    grad_scale() returns (a - a*g) + a*g evaluated in fp32 (lsq.py:6-9), which can differ from `a` in
    the last ulp.  Integer levels are defined with this effective scale; the HIP kernels reproduce it."""
    a = torch.where(alpha > 1e-5, alpha, torch.tensor(1e-5, dtype=alpha.dtype))
    t = a * gscale
    return (a - t) + t


def lsq_levels(x, alpha, lo, hi, gscale):
    """Integer codes q = RNE(clamp(x / a_eff, lo, hi)); alpha already broadcastable."""
    a = lsq_effective_scale(alpha, gscale)
    return torch.clamp(x / a, lo, hi).round().to(torch.int32)


def lsq_backward_closed_form(g, x, alpha, lo, hi, gscale):
    """Closed-form gradients of _lsq_core wrt x and the broadcast alpha.

    dx = (g*a)/a * 1[lo <= v <= hi]  (autograd's op order: mul-by-scale backward, then div backward, so dx
    equals g only up to one ulp);  d(alpha) elementwise = gscale * g * (q - v if in-range else clamp(v)).
    The caller sums the elementwise d(alpha) over the broadcast axes."""
    a = lsq_effective_scale(alpha, gscale)
    v = x / a
    inr = (v >= lo) & (v <= hi)
    u = torch.clamp(v, lo, hi)
    q = u.round()
    dx = torch.where(inr, (g * a) / a, torch.zeros_like(g))
    dalpha = gscale * g * torch.where(inr, q - v, u)
    return dx, dalpha


# ----------------------------------------------------------------------------------------------
# Q-modules
# ----------------------------------------------------------------------------------------------


def _bias_add(x, b):
    return x + b.expand_as(x)                                            # qbias.py:10


def qlinear(x, p, wbits, abits, unsigned=False):
    """QLinear.forward (qlinear.py:58-73)."""
    W, _, _ = statsq(p["weight"], wbits)                                 # :62
    x = _bias_add(x, p["move_b4.bias"])                                  # :66
    x = lsq_token(x, p["input_quant_fn.s"], abits, unsigned)             # :67
    x = _bias_add(x, p["move_aft.bias"])                                 # :68
    out = F.linear(x, W)                                                 # :69
    out = out + p["bias"].view(1, -1).expand_as(out)                     # :71
    return out


def qmlp(x, p, wbits, abits):
    """QMLP.forward (qlinear.py:123-136): fc1 signed, exact-erf GELU, fc2 unsigned (:118-120)."""
    h = qlinear(x, _sub(p, "fc1."), wbits, abits, unsigned=False)
    h = F.gelu(h)
    return qlinear(h, _sub(p, "fc2."), wbits, abits, unsigned=True)


def _sub(p, prefix):
    n = len(prefix)
    return {k[n:]: v for k, v in p.items() if k.startswith(prefix)}


def qattention(x, p, num_heads, wbits, abits, pre_softmax=None):
    """QAttention.forward, the plain (non-QKR) path (attention.py:67-105).  `pre_softmax` (Swin) maps the scaled
    scores to scores + relative-position bias (+ shift mask), swin_attention_and_mlp.py:201-221."""
    B, N, C = x.shape
    d = C // num_heads
    qkv = qlinear(x, _sub(p, "qkv."), wbits, abits)                      # :69
    qkv = _bias_add(qkv, p["move_qkv_b4.bias"])                          # :71
    qkv = qkv.reshape(B, N, 3, num_heads, d).permute(2, 0, 3, 1, 4)      # :72-74
    q, k, v = qkv[0], qkv[1], qkv[2]
    q = lsq_token(q, p["quan_a_q_fn.s"], abits, False)                   # :77
    k = lsq_token(k, p["quan_a_k_fn.s"], abits, False)                   # :78
    v = v.permute(0, 2, 1, 3).reshape(B, N, C)                           # :80
    v = lsq_channel(v, p["quan_a_v_fn.s"], abits)                        # :81
    q = q.permute(0, 2, 1, 3).reshape(B, N, C)                           # :85-87
    k = k.permute(0, 2, 1, 3).reshape(B, N, C)
    q = _bias_add(q, p["move_q_aft.bias"])                               # :88-90
    k = _bias_add(k, p["move_k_aft.bias"])
    v = _bias_add(v, p["move_v_aft.bias"])
    q = q.reshape(B, N, num_heads, d).permute(0, 2, 1, 3)                # :92-94
    k = k.reshape(B, N, num_heads, d).permute(0, 2, 1, 3)
    v = v.reshape(B, N, num_heads, d).permute(0, 2, 1, 3)
    attn = (q @ k.transpose(-2, -1).contiguous()) * (d ** -0.5)          # :96
    if pre_softmax is not None:
        attn = pre_softmax(attn)
    prob = F.softmax(attn, dim=-1)                                       # :97
    prob = lsq_token(prob, p["quan_a_softmax_fn.s"], abits, True)        # :99
    out = (prob @ v).transpose(1, 2).reshape(B, N, C)                    # :102
    return qlinear(out, _sub(p, "proj."), wbits, abits)                  # :103


def _rel_index(ws):
    ch, cw = torch.arange(ws[0]), torch.arange(ws[1])
    coords = torch.stack(torch.meshgrid(ch, cw, indexing="ij")).flatten(1)
    rel = (coords[:, :, None] - coords[:, None, :]).permute(1, 2, 0).contiguous()
    rel[:, :, 0] += ws[0] - 1
    rel[:, :, 1] += ws[1] - 1
    rel[:, :, 0] *= 2 * ws[1] - 1
    return rel.sum(-1).view(-1)


def swin_window_attention(x, p, num_heads, window, shift, wbits, abits, qkr):
    """QAttention_swin.forward / QAttention_swin_qkreparam.forward (swin_attention_and_mlp.py:143-251, :344-461):
    pad, cyclic shift, window partition, the DeiT attention on each window with relative-position bias and shift mask
    added to the scaled scores, reverse."""
    ws, ss = list(window), list(shift)
    N = ws[0] * ws[1]
    bias = p["relative_position_bias_table"][_rel_index(ws)].view(N, N, -1).permute(2, 0, 1).contiguous().unsqueeze(0)
    B, H, W, C = x.shape
    pad_r = (ws[1] - W % ws[1]) % ws[1]
    pad_b = (ws[0] - H % ws[0]) % ws[0]
    x = F.pad(x, (0, 0, 0, pad_r, 0, pad_b))                              # :148-150
    pH, pW = x.shape[1], x.shape[2]
    if ws[0] >= pH:
        ss[0] = 0                                                        # :153-156
    if ws[1] >= pW:
        ss[1] = 0
    if sum(ss) > 0:
        x = torch.roll(x, shifts=(-ss[0], -ss[1]), dims=(1, 2))          # :159-160
    nW = (pH // ws[0]) * (pW // ws[1])
    x = x.view(B, pH // ws[0], ws[0], pW // ws[1], ws[1], C).permute(0, 1, 3, 2, 4, 5).reshape(B * nW, N, C)   # :163-165

    def pre_softmax(attn):
        attn = attn + bias                                               # :203
        if sum(ss) > 0:                                                  # :205-221
            m = x.new_zeros((pH, pW))
            hs = ((0, -ws[0]), (-ws[0], -ss[0]), (-ss[0], None))
            wsl = ((0, -ws[1]), (-ws[1], -ss[1]), (-ss[1], None))
            count = 0
            for h in hs:
                for w in wsl:
                    m[h[0]:h[1], w[0]:w[1]] = count
                    count += 1
            m = m.view(pH // ws[0], ws[0], pW // ws[1], ws[1]).permute(0, 2, 1, 3).reshape(nW, N)
            m = m.unsqueeze(1) - m.unsqueeze(2)
            m = m.masked_fill(m != 0, float(-100.0)).masked_fill(m == 0, float(0.0))
            attn = attn.view(B, nW, num_heads, N, N) + m.unsqueeze(1).unsqueeze(0)
            attn = attn.view(-1, num_heads, N, N)
        return attn

    fn = qattention_qkr if qkr else qattention
    y = fn(x, {k: v for k, v in p.items() if not k.startswith("relative_position")}, num_heads, wbits, abits, pre_softmax)
    y = y.view(B, pH // ws[0], pW // ws[1], ws[0], ws[1], C).permute(0, 1, 3, 2, 4, 5).reshape(B, pH, pW, C)   # :231-232
    if sum(ss) > 0:
        y = torch.roll(y, shifts=(ss[0], ss[1]), dims=(1, 2))            # :235-236
    return y[:, :H, :W, :].contiguous()                                  # :239


def swin_patch_merging(x, p, wbits, abits, eps=1e-5):
    """PatchMerging.forward with a quantised `reduction` (swin.py:40-60); QLinear on a 4-D input: the LSQ step is
    indexed by the feature-map column (x.shape[-2])."""
    H, W, _ = x.shape[-3:]
    x = F.pad(x, (0, 0, 0, W % 2, 0, H % 2))
    x = torch.cat([x[..., 0::2, 0::2, :], x[..., 1::2, 0::2, :], x[..., 0::2, 1::2, :], x[..., 1::2, 1::2, :]], -1)
    x = F.layer_norm(x, (x.shape[-1],), p["norm.weight"], p["norm.bias"], eps)
    return qlinear(x, _sub(p, "reduction."), wbits, abits)


def swin_forward(img, sd, cfg):
    """SwinTransformer.forward (swin.py:441-470) with quantised stem, window attention, MLP, reductions and head.
    cfg: dict(depths, num_heads, window, patch, wbits, abits, qkr)."""
    wb, ab = cfg["wbits"], cfg["abits"]
    eps = 1e-5
    x = qconv_patch_embed_nhwc(img, _sub(sd, "features.0.0."), cfg["patch"])                  # features.0.0 + Permute
    x = F.layer_norm(x, (x.shape[-1],), sd["features.0.2.weight"], sd["features.0.2.bias"], eps)
    fi = 1
    for si, depth in enumerate(cfg["depths"]):
        for li in range(depth):
            pre = "features.%d.%d." % (fi, li)
            C = x.shape[-1]
            shift = [0 if li % 2 == 0 else w // 2 for w in cfg["window"]]
            h = F.layer_norm(x, (C,), sd[pre + "norm1.weight"], sd[pre + "norm1.bias"], eps)
            x = x + swin_window_attention(h, _sub(sd, pre + "attn."), cfg["num_heads"][si], cfg["window"], shift, wb, ab,
                                          cfg["qkr"])
            h = F.layer_norm(x, (C,), sd[pre + "norm2.weight"], sd[pre + "norm2.bias"], eps)
            x = x + qmlp(h, _sub(sd, pre + "mlp."), wb, ab)
        fi += 1
        if si < len(cfg["depths"]) - 1:
            x = swin_patch_merging(x, _sub(sd, "features.%d." % fi), wb, ab)
            fi += 1
    x = F.layer_norm(x, (x.shape[-1],), sd["norm.weight"], sd["norm.bias"], eps)
    x = x.permute(0, 3, 1, 2).mean(dim=(2, 3))                             # AdaptiveAvgPool2d(1) + flatten
    return qhead(x, _sub(sd, "head."))


def qattention_qkr(x, p, num_heads, wbits, abits, pre_softmax=None):
    """QAttention_qkreparam.forward (attention.py:174-222); the `_4_cga` twin (:291-339) is
    numerically identical in value and gradient (SURVEY.md §7 hard part 9)."""
    B, N, C = x.shape
    H = num_heads
    d = C // H
    xq = _bias_add(x, p["quant_x_4_qkv.move_b4.bias"])                   # :177 -> qlinear.py:21-26
    xq = lsq_token(xq, p["quant_x_4_qkv.input_quant_fn.s"], abits, False)
    xq = _bias_add(xq, p["quant_x_4_qkv.move_aft.bias"])
    Wv, _, _ = statsq(p["v.weight"], wbits)                              # :179
    v = F.linear(xq, Wv)                                                 # :180
    v = v + p["v.bias"].view(1, -1).expand_as(v)                         # :181
    v = _bias_add(v, p["move_v_b4.bias"])                                # :184
    v = lsq_channel(v, p["quan_a_v_fn.s"], abits)                        # :185
    v = _bias_add(v, p["move_v_aft.bias"])                               # :186
    v = v.reshape(B, N, H, d).permute(0, 2, 1, 3)                        # :187
    Wq = p["q.weight"].reshape(H, d, C)                                  # :190
    Wk = p["k.weight"].reshape(H, d, C)                                  # :191
    Wqk = Wq.transpose(-2, -1).contiguous() @ Wk                         # :193  (H, C, C)
    Wqk = Wqk.reshape(H * C, C)                                          # :194
    Wqk_q, _, _ = statsq(Wqk, wbits)                                     # :195
    Wqk_q = Wqk_q.reshape(H, C, C)                                       # :196
    qkx = torch.einsum("HDC,BCN->BHDN", Wqk_q, xq.transpose(-2, -1).contiguous())  # :200
    qkx = qkx.permute(0, 3, 1, 2).reshape(B, N, H * C)                   # :201
    qkx = _bias_add(qkx, p["move_qkx_b4.bias"])                          # :202
    qkx = qkx.reshape(B, N * H, C)                                       # :203
    qkx = lsq_token(qkx, p["quan_a_qkx_fn.s"], abits, False)             # :204  s per (token, head)
    qkx = qkx.reshape(B, N, H * C)                                       # :205
    qkx = _bias_add(qkx, p["move_qkx_aft.bias"])                         # :206
    qkx = qkx.reshape(B, N, H, -1).permute(0, 2, 3, 1)                   # :207  (B,H,C,N)
    attn = torch.einsum("BNC,BHCD->BHND", xq, qkx)                       # :210
    attn = attn * (d ** -0.5)                                            # :213
    if pre_softmax is not None:
        attn = pre_softmax(attn)
    prob = F.softmax(attn, dim=-1)                                       # :214
    prob = lsq_token(prob, p["quan_a_softmax_fn.s"], abits, True)        # :216
    out = (prob @ v).transpose(1, 2).reshape(B, N, C)                    # :219
    return qlinear(out, _sub(p, "proj."), wbits, abits)                  # :220


def qconv_patch_embed(img, p, patch):
    """LSQ_QConv2d.forward (qlinear.py:166-177) + timm PatchEmbed flatten/transpose; W8A8."""
    signed = bool(p["input_quant_fn.signed"].item() != 0)
    W = lsq_convw(p["weight"], p["lsqw_fn.s"])                           # :168
    hh, ww = img.shape[-2], img.shape[-1]
    x = img + p["move_b4.bias"].reshape(ww, hh).expand_as(img)           # :171, qbias.py:21
    x = lsq_img(x, p["input_quant_fn.s"], signed)                        # :172
    x = x + p["move_aft.bias"].reshape(ww, hh).expand_as(x)              # :173
    y = F.conv2d(x, W, p["bias"], stride=patch)                          # :174
    return y.flatten(2).transpose(1, 2)


def qconv_patch_embed_nhwc(img, p, patch):
    """Swin stem: LSQ_QConv2d then Permute([0,2,3,1]) (swin.py:385-393)."""
    B = img.shape[0]
    y = qconv_patch_embed(img, p, patch)                                 # (B, gh*gw, C), row-major over (gh, gw)
    g = img.shape[-1] // patch
    return y.reshape(B, img.shape[-2] // patch, g, -1)


def qhead(x, p):
    """LSQ_QLinear4head.forward (qlinear.py:223-238); W8A8, per-tensor input scale."""
    W = lsq_roww(p["weight"], p["lsqw_fn.s"])                            # :227
    x = _bias_add(x, p["move_b4.bias"])                                  # :231
    x = lsq_tensor(x, p["input_quant_fn.s"])                             # :232
    x = _bias_add(x, p["move_aft.bias"])                                 # :233
    out = F.linear(x, W)                                                 # :234
    return out + p["bias"].view(1, -1).expand_as(out)                    # :236


# ----------------------------------------------------------------------------------------------
# DeiT (distilled) forward (deit.py:32-67; deit_vision_transformer.py:154-164)
# ----------------------------------------------------------------------------------------------


def deit_forward(img, sd, cfg, training=True):
    """cfg: dict(depth, num_heads, patch, wbits, abits, qkr: bool, ln_eps).  sd: flat state dict with the
    reference key names.  Returns (cls_logits, dist_logits) in training mode, their mean otherwise."""
    H = cfg["num_heads"]
    wb, ab = cfg["wbits"], cfg["abits"]
    eps = cfg.get("ln_eps", 1e-6)                                        # deit.py:76
    x = qconv_patch_embed(img, _sub(sd, "patch_embed.proj."), cfg["patch"])
    B = x.shape[0]
    cls = sd["cls_token"].expand(B, -1, -1)                              # deit.py:34
    dist = sd["dist_token"].expand(B, -1, -1)
    x = torch.cat((cls, dist, x), dim=1)                                 # :38
    x = x + sd["pos_embed"]                                              # :39
    C = x.shape[-1]
    attn_fn = qattention_qkr if cfg["qkr"] else qattention
    for i in range(cfg["depth"]):
        pre = "blocks.%d." % i
        h = F.layer_norm(x, (C,), sd[pre + "norm1.weight"], sd[pre + "norm1.bias"], eps)
        x = x + attn_fn(h, _sub(sd, pre + "attn."), H, wb, ab)          # dvt.py:161-162
        h = F.layer_norm(x, (C,), sd[pre + "norm2.weight"], sd[pre + "norm2.bias"], eps)
        x = x + qmlp(h, _sub(sd, pre + "mlp."), wb, ab)                  # dvt.py:163
    x = F.layer_norm(x, (C,), sd["norm.weight"], sd["norm.bias"], eps)   # deit.py:47
    cls_x = qhead(x[:, 0], _sub(sd, "head."))                            # deit.py:60
    dist_x = qhead(x[:, 1], _sub(sd, "head_dist."))
    if training:
        return cls_x, dist_x
    return (cls_x + dist_x) / 2                                          # deit.py:64


def deit_fp32_forward(img, sd, depth, num_heads, patch=16, training=True, ln_eps=1e-6):
    """The fp32 (un-quantised) distilled DeiT the KD recipes use as teacher: PatchEmbed conv, cls / dist tokens, pos_embed
    (deit.py:27-40), `depth` blocks x + attn(norm1(x)), x + mlp(norm2(x)) (deit_vision_transformer.py:85-164: qkv linear,
    softmax((q k^T) * scale), (attn @ v), proj; fc1, exact GELU, fc2), final norm, the two heads (deit.py:56-67).
    sd: state dict with the reference's key names.  Training mode returns (cls, dist), eval mode their mean."""
    x = F.conv2d(img, sd["patch_embed.proj.weight"], sd["patch_embed.proj.bias"], stride=patch)
    x = x.flatten(2).transpose(1, 2)
    B = x.shape[0]
    x = torch.cat((sd["cls_token"].expand(B, -1, -1), sd["dist_token"].expand(B, -1, -1), x), dim=1) + sd["pos_embed"]
    C = x.shape[-1]
    d = C // num_heads
    for i in range(depth):
        p = _sub(sd, "blocks.%d." % i)
        h = F.layer_norm(x, (C,), p["norm1.weight"], p["norm1.bias"], ln_eps)
        N = h.shape[1]
        qkv = F.linear(h, p["attn.qkv.weight"], p["attn.qkv.bias"]).reshape(B, N, 3, num_heads, d).permute(2, 0, 3, 1, 4)
        q, k, v = qkv[0], qkv[1], qkv[2]
        attn = ((q @ k.transpose(-2, -1)) * (d ** -0.5)).softmax(dim=-1)                 # :81-83
        a = (attn @ v).transpose(1, 2).reshape(B, N, C)
        x = x + F.linear(a, p["attn.proj.weight"], p["attn.proj.bias"])
        h = F.layer_norm(x, (C,), p["norm2.weight"], p["norm2.bias"], ln_eps)
        h = F.linear(F.gelu(F.linear(h, p["mlp.fc1.weight"], p["mlp.fc1.bias"])), p["mlp.fc2.weight"], p["mlp.fc2.bias"])
        x = x + h
    x = F.layer_norm(x, (C,), sd["norm.weight"], sd["norm.bias"], ln_eps)
    cls = F.linear(x[:, 0], sd["head.weight"], sd["head.bias"])
    dist = F.linear(x[:, 1], sd["head_dist.weight"], sd["head_dist.bias"])
    return (cls, dist) if training else (cls + dist) / 2


def kd_loss_soft_and_hard(cls_out, dist_out, hard_target, soft_target):
    """KDLossSoftandHard.forward (quantization/utils.py:59-77) with KLLossSoft (:44-57), T = 1."""
    tp = F.softmax(soft_target, dim=1)
    lp = F.log_softmax(dist_out, dim=1)
    soft = (-torch.sum(tp * lp, dim=1)).mean()
    hard = F.cross_entropy(cls_out, hard_target)
    return soft + hard


# ----------------------------------------------------------------------------------------------
# CGA (cga.py:450-469, :953-1013)
# ----------------------------------------------------------------------------------------------


def cga_freeze_idx(W, bits, boundary_range=0.005, s=None):
    """freeze_outside_boundary_weight_idx (cga.py:450-469): 1.0 where the weight is frozen.
    s: test hook -- the per-row scale to use instead of 2 * mean|W| (tests inject the device's scale, which is the correctly
    rounded row mean where torch-CPU's cascade sum may be one ulp off, to state the mask's bit-exactness GIVEN the scale)."""
    W = W.detach()
    if s is None:
        s = 2 * torch.mean(W.abs(), dim=1, keepdim=True)                 # :462
    else:
        s = s.detach().reshape(-1, 1).to(W.dtype)
    c = torch.clamp(W / s, min=-1.0, max=1.0 - 1e-6)                     # :456
    n = float(2 ** (bits - 1))
    b4 = c * n - 0.5                                                     # :458
    r = torch.round(b4)
    lo_i, hi_i = int(r.min().item()), int(r.max().item())                # :460, :465
    notfrozen = torch.zeros_like(W)
    for i in range(lo_i, hi_i):                                          # np.arange(min, max) :465
        within = ((b4 - i) <= (0.5 + boundary_range)) & ((b4 - i) >= (0.5 - boundary_range))  # :466
        notfrozen = notfrozen + within.float()
    return 1.0 - notfrozen                                               # :469


def cga_mask_grad(grad, frz):
    return grad * frz * 0.0 + grad * (1 - frz)                           # cga.py:962


def cga_restore(W_new, W_old, frz):
    return W_new * (1 - frz) + (W_old * frz)                             # cga.py:964, :994-997


# ----------------------------------------------------------------------------------------------
# Input pipeline (train.py:579-629).  The arithmetic lives in timm==0.5.4 (README.md:19), which is not vendored under
# /root/reference: restated here from its published source -- timm/data/mixup.py (FastCollateMixup._mix_batch_collate,
# mixup_target, one_hot), timm/data/loader.py (PrefetchLoader.__iter__: float().sub_(mean).div_(std) with mean / std * 255)
# and timm/data/random_erasing.py (RandomErasing._erase, mode 'pixel': the rectangle is filled with normal noise).  The
# reference holds no tests or fixtures for this path (parity unpinned upstream); pinned here by hand-computed cases in
# tests/test_host_logic.py.
# ----------------------------------------------------------------------------------------------


def input_pipeline(images_u8, lam, use_cutmix, box, rects, noise, mean=(0.485, 0.456, 0.406), std=(0.229, 0.224, 0.225)):
    """images_u8: uint8 numpy [B][C][H][W]; lam: python float (1.0: no mixing); box: (yl, yh, xl, xh); rects: int [B][4]
    {top, left, h, w} (h = 0: not erased) or None; noise: float32 [B][C][H][W].  Returns float32 [B][C][H][W]."""
    import numpy as np
    B = images_u8.shape[0]
    mixed_all = np.zeros(images_u8.shape, dtype=np.uint8)
    yl, yh, xl, xh = box
    for i in range(B):                                                  # _mix_batch_collate
        j = B - i - 1
        mixed = images_u8[i]
        if lam != 1.0:
            if use_cutmix:
                mixed = mixed.copy()
                mixed[:, yl:yh, xl:xh] = images_u8[j][:, yl:yh, xl:xh]
            else:
                mixed = mixed.astype(np.float32) * np.float32(lam) + images_u8[j].astype(np.float32) * np.float32(1 - lam)
                np.rint(mixed, out=mixed)
        mixed_all[i] += mixed.astype(np.uint8)
    x = torch.from_numpy(mixed_all).float()                             # PrefetchLoader
    m = torch.tensor([v * 255 for v in mean]).view(1, -1, 1, 1)
    sd = torch.tensor([v * 255 for v in std]).view(1, -1, 1, 1)
    x = x.sub_(m).div_(sd)
    if rects is not None:                                               # RandomErasing._erase
        nz = torch.as_tensor(noise)
        for i in range(B):
            top, left, h, w = [int(v) for v in rects[i]]
            if h > 0:
                x[i, :, top:top + h, left:left + w] = nz[i, :, top:top + h, left:left + w]
    return x


def mixup_target(target, num_classes, lam, smoothing):
    """timm.data.mixup.mixup_target (device='cpu')."""
    off_value = smoothing / num_classes
    on_value = 1.0 - smoothing + off_value
    t = target.long().view(-1, 1)
    y1 = torch.full((t.shape[0], num_classes), off_value).scatter_(1, t, on_value)
    y2 = torch.full((t.shape[0], num_classes), off_value).scatter_(1, t.flip(0), on_value)
    return y1 * lam + y2 * (1.0 - lam)
