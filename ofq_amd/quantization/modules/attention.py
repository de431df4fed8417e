"""QAttention / QAttention_qkreparam / QAttention_qkreparam_4_cga — drop-ins for
src/quantization/modules/attention.py (:12, :107, :224): same constructor signatures, parameter names
and `forward(x[B,N,C]) -> (y[B,N,C], None)`.

Data layout (MI355X-first, not the reference's permute/contiguous chain): activations stay token-major
(B, N, C) end to end; heads are column slices addressed through GEMM strides; the attention matrices are
(B, H, N, Np) with Np = N rounded to 4 so rows stay float4-aligned.  No transposes are materialised."""
import torch
import torch.nn as nn

from .qbias import LearnableBias
from .qlinear import QLinear, LSQ_input
from ..quantizer.lsq import LsqQuantizer, LsqQuantizer4v
from ..quantizer.statsq import StatsQuantizer, StatsQuantizer_specific_4_qkreparam_cga
from ...deit_vision_transformer import Attention as deit_attention
from ... import ops
from ...functional import (LinearFn, WqkFn, QKRScoresFn, QKScoresFn, SoftmaxLsqFn, PVFn, QKVSplitLsqFn, codes_linear,
                           codes_linear_ok, QKRScoresCodesFn, SoftmaxLsqCodesFn, PVCodesFn, QKVSplitLsqCodesFn,
                           QKScoresCodesFn, ScoresSoftmaxCodesFn, scores_softmax_fusable)
from . import qlinear as _ql


def _qlinear_kwargs(weight_bits, input_bits, weight_channelwise, input_channelwise, weight_quant_method,
                    input_quant_method, aq_learnable, wq_learnable, pretrained_initialized):
    return dict(weight_bits=weight_bits, input_bits=input_bits, weight_channelwise=weight_channelwise,
                input_channelwise=input_channelwise, weight_quant_method=weight_quant_method,
                input_quant_method=input_quant_method, aq_learnable=aq_learnable, wq_learnable=wq_learnable,
                symmetric=True, pretrained_initialized=pretrained_initialized)


def _softmax_init(quant, S, N, alpha, addend):
    """lazily initialise the softmax quantiser's step like lsq.py:544-569 (data-dependent, first batch)."""
    if not quant.initialized_alpha or quant.s is None:
        with torch.no_grad():
            a = S[..., :N].detach() * alpha
            if addend is not None:
                B, H = a.shape[0], a.shape[1]
                P = addend.shape[0]
                a = (a.reshape(B * H // P, P, N, N) + addend[..., :N].detach()).reshape(B, H, N, N)
            quant.init_from(torch.softmax(a, dim=-1))


def _fit_addend(addend, S):
    """pad the (P, N, N) additive term to the row stride of the score tensor"""
    if addend is not None and addend.shape[-1] != S.shape[-1]:
        addend = torch.nn.functional.pad(addend, (0, S.shape[-1] - addend.shape[-1]))
    return None if addend is None else addend.contiguous()


def _pad_addend(addend, N):
    """the (P, N, N) additive term padded to the score row stride the code kernels use (N rounded up to 16)"""
    if addend is None:
        return None
    Np = (N + 15) // 16 * 16
    if addend.shape[-1] != Np:
        addend = torch.nn.functional.pad(addend, (0, Np - addend.shape[-1]))
    return addend.contiguous()


def _softmax_lsq(quant, S, N, alpha, addend=None):
    """unsigned per-query-token LSQ on softmax(S*alpha [+ addend])."""
    addend = _fit_addend(addend, S)
    _softmax_init(quant, S, N, alpha, addend)
    return SoftmaxLsqFn.apply(S, quant.s, N, alpha, quant.thd_pos, addend)


class QAttention(deit_attention):
    """Plain (non-QKR) quantised attention, attention.py:12-105."""

    def __init__(self, m: deit_attention, weight_bits=8, input_bits=8, aq_learnable=True, wq_learnable=True,
                 weight_channelwise=True, input_channelwise=True, weight_quant_method="statsq",
                 input_quant_method="lsq", pretrained_initialized=False, **kwargs):
        assert type(m) == deit_attention
        if m.attn_drop.p != 0:
            raise ValueError("attention dropout is 0 in every OFQ recipe; the quantised attention core does not apply it "
                             "(the reference applies attn_drop after the softmax quantiser, attention.py:100 / :217)")
        # Like the reference (attention.py:18-24) the base class is built WITHOUT qkv_bias, and the two QLinear wrappers
        # below take `self.qkv` / `self.proj` -- the layers this constructor has just created -- not `m.qkv` / `m.proj`
        # (attention.py:29-30, :42-43): with pretrained_initialized the qkv / proj weights of a QAttention are therefore
        # the fresh default initialisation (and qkv.bias the QLinear's own), NOT the source module's.  Reproduced, with the
        # same order of random draws, so that a seeded construction gives the reference's parameters bit for bit
        # (tests/golden g10); checkpoints loaded afterwards (train.py:515-516) override them either way.
        super().__init__(dim=m.qkv.in_features, num_heads=m.num_heads, attn_drop=m.attn_drop.p,
                         proj_drop=m.proj_drop.p, qqkkvv=m.qqkkvv)
        self.weight_bits = weight_bits
        self.input_bits = input_bits
        self.input_channelwise = input_channelwise
        if input_bits >= 32:
            raise ValueError("QAttention: input_bits >= 32 (no activation quantisation) is not on the hot path")
        kw = _qlinear_kwargs(weight_bits, input_bits, weight_channelwise, input_channelwise, weight_quant_method,
                             input_quant_method, aq_learnable, wq_learnable, pretrained_initialized)
        self.qkv = QLinear(m=self.qkv, **kw)
        self.proj = QLinear(m=self.proj, **kw)
        self.quan_a_q_fn = LsqQuantizer(bit=input_bits, all_positive=False, per_channel=True, learnable=aq_learnable)
        self.quan_a_k_fn = LsqQuantizer(bit=input_bits, all_positive=False, per_channel=True, learnable=aq_learnable)
        self.quan_a_v_fn = LsqQuantizer4v(bit=input_bits, all_positive=False, per_channel=True, learnable=aq_learnable)
        C = m.qkv.in_features
        self.move_qkv_b4 = LearnableBias(C * 3)
        self.move_q_aft = LearnableBias(C)
        self.move_k_aft = LearnableBias(C)
        self.move_v_aft = LearnableBias(C)
        self.quan_a_softmax_fn = LsqQuantizer(bit=input_bits, all_positive=True, per_channel=True, learnable=aq_learnable)

    def fused_input_quant(self, in_shape):
        """The qkv projection's input quantiser, for a producer that can apply it itself (LayerNorm + LSQ in one kernel,
        deit_vision_transformer.Block.forward_fused); None when qkv needs the fp32 values."""
        return self.qkv.fused_input_quant(in_shape)

    def forward(self, x, pre_quant=None):
        return self.proj_drop(self.proj(plain_attention_core(self, x, self.scale, pre_quant=pre_quant))), None   # :103-105


def _plain_lazy_init(self, qkv):
    B, N, C3 = qkv.shape
    C = C3 // 3
    todo = [(i, qz) for i, qz in enumerate((self.quan_a_q_fn, self.quan_a_k_fn, self.quan_a_v_fn))
            if not qz.initialized_alpha or qz.s is None]
    if not todo:
        return                                          # every later step: nothing to do (and no qkv-sized add)
    with torch.no_grad():
        t = qkv.detach() + self.move_qkv_b4.bias.detach()
        for i, qz in todo:
            qz.init_from(t[..., i * C:(i + 1) * C])


def plain_attention_core(self, x, scale, addend=None, pre_quant=None):
    """qkv projection -> offsets/LSQ on q,k,v -> scores -> softmax+LSQ -> P.V, everything before `proj`
    (attention.py:69-102; shared with the Swin window attention, swin_attention_and_mlp.py:173-228)."""
    if True:
        B, N, C = x.shape
        H = self.num_heads
        qkv = self.qkv(x, pre_quant=pre_quant)                                   # attention.py:69
        _plain_lazy_init(self, qkv)
        lo, hi = self.quan_a_q_fn.thd_neg, self.quan_a_q_fn.thd_pos
        gq = ops.LsqGeom(B, N, C, C, 0, lo, hi, B * C, ldx=3 * C, ldy=C)         # s per token, M = B*H*d
        gv = ops.LsqGeom(B * N, 1, C, C, 1, lo, hi, B * N, ldx=3 * C, ldy=C)     # s per channel, M = B*N
        d = C // H
        sm = self.quan_a_softmax_fn
        if _ql.USE_CODE_GEMM and _ql.PLAIN_ATTN_CODES and C % 16 == 0 and d % 16 == 0 and N <= 256 and lo >= -128 \
                and hi <= 127 and sm.thd_pos <= 127:
            # the core on the integer codes, like the QKR path: q_hat / k_hat / v_hat / P_hat exist only as int8 codes, the
            # scores and P.V are exact int8 GEMMs with the offsets as epilogue terms, the backward products bf16-split GEMMs
            q, k, v, qc, kc, vc = QKVSplitLsqCodesFn.apply(qkv, self.move_qkv_b4.bias, self.quan_a_q_fn.s,
                                                           self.quan_a_k_fn.s, self.quan_a_v_fn.s, self.move_q_aft.bias,
                                                           self.move_k_aft.bias, self.move_v_aft.bias, gq, gq, gv)   # :71-90
            link = {}
            saux = {"qcodes": qc, "kcodes": kc, "sq": self.quan_a_q_fn.s.detach(), "gq": gq.gscale,
                    "sk": self.quan_a_k_fn.s.detach(), "gk": gq.gscale, "bq": self.move_q_aft.bias.detach(),
                    "bk": self.move_k_aft.bias.detach(), "H": H, "link": link, "plain_pre": getattr(self, "_plain_pre", None)}
            pv_aux = {"vcodes": vc, "sv": self.quan_a_v_fn.s.detach(), "gv": gv.gscale, "bav": self.move_v_aft.bias.detach()}
            if scores_softmax_fusable(N) and sm.initialized_alpha and sm.s is not None:
                saux.update(plain=True, alpha=scale, hi=sm.thd_pos, vlink=pv_aux)      # (vlink: dP GEMM + softmax backward fused)
                P, pcodes, rp = ScoresSoftmaxCodesFn.apply(q, k, sm.s, saux, _pad_addend(addend, N))                 # :96-99
            else:
                S = QKScoresCodesFn.apply(q, k, saux)                                                                 # :96
                addend = _fit_addend(addend, S)
                _softmax_init(sm, S, N, scale, addend)
                P, pcodes, rp = SoftmaxLsqCodesFn.apply(S, sm.s, N, scale, sm.thd_pos, link, addend)                  # :97-99
            gp = 1.0 / (sm.thd_pos * B * H * N) ** 0.5
            pv_aux.update(pcodes=pcodes, rp=rp, sp=sm.s.detach(), gp=gp)
            return PVCodesFn.apply(P, v, pv_aux)                                                                      # :102
        q, k, v = QKVSplitLsqFn.apply(qkv, self.move_qkv_b4.bias, self.quan_a_q_fn.s, self.quan_a_k_fn.s,
                                      self.quan_a_v_fn.s, self.move_q_aft.bias, self.move_k_aft.bias,
                                      self.move_v_aft.bias, gq, gq, gv)          # :71-90
        S = QKScoresFn.apply(q, k, H)                                            # :96 (scale folded into softmax)
        P = _softmax_lsq(self.quan_a_softmax_fn, S, N, scale, addend)            # :97-99
        return PVFn.apply(P, v, N)                                               # :102


class QAttention_qkreparam(deit_attention):
    """Query-key reparameterised attention, attention.py:107-222: StatsQ acts on W_q^T W_k per head."""

    _qk_quant_cls = StatsQuantizer

    def __init__(self, m: deit_attention, weight_bits=8, input_bits=8, aq_learnable=True, wq_learnable=True,
                 weight_channelwise=True, input_channelwise=True, weight_quant_method="statsq",
                 input_quant_method="lsq", pretrained_initialized=False, boundaryRange=0.005, **kwargs):
        assert type(m) == deit_attention
        if m.attn_drop.p != 0:
            raise ValueError("attention dropout is 0 in every OFQ recipe; the quantised attention core does not apply it "
                             "(the reference applies attn_drop after the softmax quantiser, attention.py:100 / :217)")
        # base class without qkv_bias and proj wrapped from `self.proj`, as the reference does (attention.py:113-119,
        # :143-144): q / k / v are copied from the source module, proj keeps this constructor's fresh initialisation
        super().__init__(dim=m.qkv.in_features, num_heads=m.num_heads, attn_drop=m.attn_drop.p,
                         proj_drop=m.proj_drop.p, qqkkvv=m.qqkkvv)
        self.weight_bits = weight_bits
        self.input_bits = input_bits
        self.input_channelwise = input_channelwise
        C = m.qkv.in_features
        self.quant_x_4_qkv = LSQ_input(bit=input_bits, all_positive=False, learnable=aq_learnable, learanbaleBiasdim=C)
        self.q = nn.Linear(C, C, bias=False)
        self.k = nn.Linear(C, C, bias=False)
        self.v = nn.Linear(C, C)
        if pretrained_initialized:
            with torch.no_grad():
                w, b = m.qkv.weight.detach(), m.qkv.bias.detach()
                self.q.weight.copy_(w[:C])
                self.k.weight.copy_(w[C:2 * C])
                self.v.weight.copy_(w[2 * C:3 * C])
                self.v.bias.copy_(b[2 * C:3 * C])
        self.qk_quant = self._make_qk_quant(wq_learnable, boundaryRange)
        self.v_quant = StatsQuantizer(num_bits=self.weight_bits, clip_learnable=wq_learnable)
        self.proj = QLinear(m=self.proj, **_qlinear_kwargs(weight_bits, input_bits, weight_channelwise, input_channelwise,
                                                        weight_quant_method, input_quant_method, aq_learnable,
                                                        wq_learnable, pretrained_initialized))
        self.quan_a_qkx_fn = LsqQuantizer(bit=input_bits, all_positive=False, per_channel=True, learnable=aq_learnable)
        self.quan_a_v_fn = LsqQuantizer4v(bit=input_bits, all_positive=False, per_channel=True, learnable=aq_learnable)
        self.move_qkx_b4 = LearnableBias(self.num_heads * C)
        self.move_qkx_aft = LearnableBias(self.num_heads * C)
        self.move_v_b4 = LearnableBias(C)
        self.move_v_aft = LearnableBias(C)
        self.quan_a_softmax_fn = LsqQuantizer(bit=input_bits, all_positive=True, per_channel=True, learnable=aq_learnable)
        del self.qkv                                                             # attention.py:172

    def _make_qk_quant(self, wq_learnable, boundaryRange):
        return StatsQuantizer(num_bits=self.weight_bits, clip_learnable=wq_learnable)

    def fused_input_quant(self, in_shape):
        """quant_x_4_qkv for a producer that applies it itself (LayerNorm + LSQ in one kernel); None when the attention
        core needs the fp32 x_hat."""
        B, N, C = in_shape
        xin = self.quant_x_4_qkv
        if not (_ql.FUSE_NORM_QUANT and _attn_codes_ok(self, N, C)):
            return None
        return {"quant": xin.input_quant_fn, "b4": xin.move_b4.bias, "baft": xin.move_aft.bias}

    def forward(self, x, pre_quant=None):
        return self.proj_drop(self.proj(qkr_attention_core(self, x, self.scale, pre_quant=pre_quant))), None   # :220-222


def _attn_codes_ok(self, N, C):
    """The whole attention core can run on integer codes (no consumer reads fp32 x_hat / v_hat / qkx_hat)."""
    H = self.num_heads
    d = C // H
    use_codes = _ql.USE_CODE_GEMM and codes_linear_ok(C, self.v_quant, self.quant_x_4_qkv.input_quant_fn)
    return (use_codes and C % 16 == 0 and d % 8 == 0 and N <= 256 and self.quan_a_softmax_fn.thd_pos <= 127
            and self.quan_a_v_fn.thd_neg >= -128 and self.quan_a_qkx_fn.thd_neg >= -128)


def qkr_attention_core(self, x, scale, addend=None, pre_quant=None):
    """Everything of QAttention_qkreparam.forward before `proj` (attention.py:177-219); also the core of the Swin
    QKR window attention (swin_attention_and_mlp.py:374-423), which adds `addend` before the softmax.
    pre_quant: (x_hat carrier, codes, geom) when the producer already applied quant_x_4_qkv (norm_quant)."""
    if True:
        B, N, C = x.shape
        H = self.num_heads
        xin = self.quant_x_4_qkv
        use_codes = _ql.USE_CODE_GEMM and codes_linear_ok(C, self.v_quant, xin.input_quant_fn)
        d = C // H
        attn_codes = _attn_codes_ok(self, N, C)
        if use_codes:
            # with the attention core on codes too, no consumer reads the fp32 x_hat / v_hat / qkx_hat values
            if pre_quant is not None:
                xq, xcodes, xgeom = pre_quant
            else:
                xq, xcodes, xgeom = xin(x, want_codes=True, need_values=not attn_codes)   # attention.py:177
            # the v / qkx quantisers are applied by the epilogue of the GEMM that produces their input (same codes)
            fuse_ok = attn_codes and _ql.FUSE_NEXT_CODES
            vspec = self.quan_a_v_fn.fusable((B, N, C), self.move_v_b4.bias, 0) if fuse_ok else None
            qspec = self.quan_a_qkx_fn.fusable((B, N * H, C), self.move_qkx_b4.bias, 0) if fuse_ok else None
            # v / qkx leave their GEMMs as codes only where qlinear.RECOMPUTE_SITES says so (backward recomputes them)
            if vspec is not None and "v" in _ql.RECOMPUTE_SITES:
                vspec["store_y"] = False
            if qspec is not None and "qkx" in _ql.RECOMPUTE_SITES:
                qspec["store_y"] = False
            # x_hat has three consumers (v GEMM, W_qk GEMM, scores): their backward passes accumulate into one buffer
            # (lin_total: the two linear layers among them run their input gradients as one GEMM, functional.CodesLinearFn)
            xacc = {"lin_total": 2} if (attn_codes and torch.is_grad_enabled()) else None
            v = codes_linear(xq, xcodes, xgeom, xin.input_quant_fn, xin.move_aft.bias, self.v.weight, self.v_quant,
                             self.v.bias, fuse=vspec, xgrad_acc=xacc)            # :179-181
        else:
            xq = xin(x)
            v = LinearFn.apply(xq, self.v_quant(self.v.weight), self.v.bias)
        if attn_codes:
            v, vcodes, vgeom = self.quan_a_v_fn.quant(v, self.move_v_b4.bias, self.move_v_aft.bias, want_codes=True,
                                                      need_values=False,
                                                      pre_codes=None if vspec is None else vspec.get("codes_out"), fused=vspec)
        else:
            v = self.quan_a_v_fn.quant(v, self.move_v_b4.bias, self.move_v_aft.bias)
        # ---- QK branch (:190-207): W_qk = per-head W_q^T W_k, StatsQ over its H*C rows
        Wqk_fp = getattr(self, "_wqk_pre", None)          # all blocks' W_qk in one batched GEMM (functional.all_wqk)
        if Wqk_fp is None:
            Wqk_fp = WqkFn.apply(self.q.weight, self.k.weight, H)
            Wqk_fp._ofq_flushes = True      # WqkFn.backward flushes the dW queue before it reads this tensor's gradient
        if use_codes:
            qkx = codes_linear(xq, xcodes, xgeom, xin.input_quant_fn, xin.move_aft.bias, Wqk_fp, self.qk_quant, None,
                               fuse=qspec, xgrad_acc=xacc)
        else:
            qkx = LinearFn.apply(xq, self.qk_quant(Wqk_fp), None)                # (B, N, H*C)   einsum :200
        if not attn_codes:
            qkx = self.quan_a_qkx_fn.quant(qkx, self.move_qkx_b4.bias, self.move_qkx_aft.bias,
                                           shape=(B, N * H, C), out_shape=(B, N, H, C))   # :201-206, s per (token, head)
            S = QKRScoresFn.apply(xq, qkx, H)                                    # :210
            P = _softmax_lsq(self.quan_a_softmax_fn, S, N, scale, addend)        # :213-216
            out = PVFn.apply(P, v, N)                                            # :219
        else:
            qkx, qcodes, qgeom = self.quan_a_qkx_fn.quant(qkx, self.move_qkx_b4.bias, self.move_qkx_aft.bias,
                                                          shape=(B, N * H, C), out_shape=(B, N, H, C), want_codes=True,
                                                          need_values=False,
                                                          pre_codes=None if qspec is None else qspec.get("codes_out"),
                                                          fused=qspec)
            link = {}
            saux = {"xcodes": xcodes, "qcodes": qcodes, "sx": xin.input_quant_fn.s.detach(), "gx": xgeom.gscale,
                    "sq": self.quan_a_qkx_fn.s.detach(), "gq": qgeom.gscale, "bax": xin.move_aft.bias.detach(),
                    "baq": self.move_qkx_aft.bias.detach(), "H": H, "link": link, "xgrad_acc": xacc}
            sm = self.quan_a_softmax_fn
            pv_aux = {"vcodes": vcodes}
            if scores_softmax_fusable(N) and sm.initialized_alpha and sm.s is not None:
                saux.update(plain=False, alpha=scale, hi=sm.thd_pos, vlink=pv_aux)
                P, pcodes, rp = ScoresSoftmaxCodesFn.apply(xq, qkx, sm.s, saux, _pad_addend(addend, N))   # :210-216
            else:
                S = QKRScoresCodesFn.apply(xq, qkx, saux)                                                  # :210
                addend = _fit_addend(addend, S)
                _softmax_init(sm, S, N, scale, addend)
                P, pcodes, rp = SoftmaxLsqCodesFn.apply(S, sm.s, N, scale, sm.thd_pos, link, addend)       # :213-216
            gp = 1.0 / (sm.thd_pos * B * H * N) ** 0.5
            pv_aux.update(pcodes=pcodes, rp=rp, sp=sm.s.detach(), gp=gp, sv=self.quan_a_v_fn.s.detach(), gv=vgeom.gscale,
                          bav=self.move_v_aft.bias.detach())
            out = PVCodesFn.apply(P, v, pv_aux)                                                              # :219
        return out


class QAttention_qkreparam_4_cga(QAttention_qkreparam):
    """attention.py:224-339.  Identical maths (its StatsQuantizer_specific_4_qkreparam_cga is value- and
    gradient-equal to StatsQuantizer); kept as its own class for `qk_reparam_type=1` and checkpoints."""

    def _make_qk_quant(self, wq_learnable, boundaryRange):
        return StatsQuantizer_specific_4_qkreparam_cga(num_bits=self.weight_bits, clip_learnable=wq_learnable,
                                                       boundaryRange=boundaryRange)
