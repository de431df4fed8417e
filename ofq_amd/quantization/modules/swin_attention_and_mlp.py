"""QMLP_swin / QAttention_swin / QAttention_swin_qkreparam(_4_cga) — drop-ins for
src/quantization/modules/swin_attention_and_mlp.py (:24, :65, :253, :463): same constructor signatures, parameter
names and `forward(x[B,H,W,C]) -> (y[B,H,W,C], None)`.

The window attention is the DeiT attention core run on (B*num_windows, 49, C) windows; the only Swin-specific
arithmetic — relative-position bias and the shifted-window mask added to the scaled scores before the softmax
(:201-221) — is one additive tensor handed to the fused softmax+LSQ kernel (`addend` of ofq_softmax_lsq_fwd).
Pad / roll / window partition and their inverses are data movement on the host (torch views and copies)."""
import os

import torch
import torch.nn as nn

from .qbias import LearnableBias
from .qlinear import QLinear, LSQ_input
from . import qlinear as _ql
from .attention import _qlinear_kwargs, plain_attention_core, qkr_attention_core, _attn_codes_ok
from ..quantizer.lsq import LsqQuantizer, LsqQuantizer4v
from ..quantizer.statsq import StatsQuantizer, StatsQuantizer_specific_4_qkreparam_cga
from ...swin import ShiftedWindowAttention, WindowGeometry, MLP as swin_MLP, relative_position_index


# round 6: the Swin MLP takes the fusions the DeiT MLP has had since round 2 (norm2 + fc1's input quantiser in one kernel each way,
# GELU + fc2's input quantiser in fc1's GEMM epilogue); OFQ_NO_SWIN_MLP_FUSE=1 is the A/B switch
SWIN_MLP_FUSE = os.environ.get("OFQ_NO_SWIN_MLP_FUSE") is None
# ... and the shifted-window partition / reverse ride in the LayerNorm passes around the attention (norm1 + partition + the attention's
# input quantiser in one kernel each way; the window-major proj output read by norm2's pass): OFQ_NO_SWIN_ATTN_FUSE=1 switches it off
SWIN_ATTN_FUSE = os.environ.get("OFQ_NO_SWIN_ATTN_FUSE") is None


class QMLP_swin(torch.nn.Module):
    """swin_attention_and_mlp.py:24-63.  Inputs are 4-D (B, H, W, C): the per-"token" LSQ step is indexed by the
    feature-map column W (x.shape[-2]), SURVEY.md Appendix A."""

    def __init__(self, *kargs, m: swin_MLP, weight_bits=8, input_bits=8, aq_learnable=True, wq_learnable=True,
                 weight_channelwise=True, input_channelwise=True, weight_quant_method="statsq", input_quant_method="lsq",
                 act_layer=nn.GELU, pretrained_initialized=False, **kwargs):
        super().__init__()
        common = dict(weight_bits=weight_bits, input_bits=input_bits, aq_learnable=aq_learnable, wq_learnable=wq_learnable,
                      weight_channelwise=weight_channelwise, input_channelwise=input_channelwise,
                      weight_quant_method=weight_quant_method, input_quant_method=input_quant_method,
                      pretrained_initialized=pretrained_initialized)
        self.fc1 = QLinear(m=m[0], symmetric=True, **common)
        self.act_layer = act_layer
        if act_layer == "rprelu":
            raise ValueError("rprelu is not part of any shipped OFQ recipe")
        self.act = act_layer() if act_layer != "None" else nn.Identity()
        self.drop1 = m[2]
        self.fc2 = QLinear(m=m[3], symmetric=False, **common)
        self.drop2 = m[4]
        self._fuse_gelu = isinstance(self.act, nn.GELU) and getattr(self.act, "approximate", "none") == "none" \
            and self.drop1.p == 0
        self.fc2._prologue = 1 if self._fuse_gelu else 0

    def fused_input_quant(self, in_shape):
        """fc1's input quantiser for a producer that can apply it itself (LayerNorm + LSQ in one kernel: SwinTransformerBlock
        .forward_fused); the 4-D geometry is the kernels' [outer = B * Hf][S = Wf][inner = C] view like any other."""
        return self.fc1.fused_input_quant(in_shape) if SWIN_MLP_FUSE else None

    def forward(self, x, pre_quant=None):
        if SWIN_MLP_FUSE and _ql.FUSE_NEXT_CODES and _ql.FUSE_NEXT_CODES_MLP and self._fuse_gelu and self.fc1.code_path():
            # as QMLP.forward (qlinear.py): fc1's GEMM epilogue applies GELU + fc2's offset and LSQ and emits fc2's input codes,
            # so the fp32 activation is written once (for the backward) and never re-read by a quantiser launch
            spec = self.fc2.input_fuse_spec(tuple(x.shape[:-1]) + (self.fc1.out_features,))
            h = self.fc1(x, fuse_next=spec, pre_quant=pre_quant)
            return self.drop2(self.fc2(h, pre_codes=None if spec is None else spec.get("codes_out"), fused=spec))
        x = self.fc1(x, pre_quant=pre_quant)
        if not self._fuse_gelu:
            x = self.drop1(self.act(x))
        return self.drop2(self.fc2(x))


class _SwinQBase(ShiftedWindowAttention):
    def _base_init(self, m, weight_bits, input_bits, input_channelwise):
        assert type(m) == ShiftedWindowAttention
        if m.attention_dropout != 0 or m.dropout != 0:
            raise ValueError("dropout is 0 in every OFQ Swin recipe; the quantised window attention does not apply it")
        ShiftedWindowAttention.__init__(self, dim=m.dim, window_size=m.window_size, shift_size=m.shift_size,
                                        num_heads=m.num_heads, qkv_bias=True, proj_bias=True, attention_dropout=0.0,
                                        dropout=0.0, qqkkvv=m.qqkkvv)
        self.weight_bits = weight_bits
        self.input_bits = input_bits
        self.input_channelwise = input_channelwise
        self.attention_dropout = 0.0
        self.dropout = 0.0
        # like the reference (:129-141) the Q-module starts from a FRESH trunc-normal bias table: the source module's
        # table is not copied, even with pretrained_initialized (ShiftedWindowAttention.__init__ above created it); the
        # same holds for proj (and qkv on the plain path): the QLinear wrappers take self.proj / self.qkv (:89-90, :297-298)

    def _window_forward(self, x, core):
        g = WindowGeometry(x, self.window_size, self.shift_size)
        self.shift_size = g.ss                                   # the reference zeroes it in place as well (:153-156)
        xw = g.partition(x)                                      # (B*nW, N, C)
        add = g.addend(self.relative_position_bias_table, self.relative_position_index, self.num_heads)
        out = core(self, xw, (self.dim // self.num_heads) ** -0.5, add)
        return g.reverse(self.proj(out)), None

    # ---- round 6: partition and reverse folded into the LayerNorm passes around the attention (SwinTransformerBlock.forward_fused)
    def _input_spec(self, win_shape):
        """The input quantiser of the window attention ({"quant", "b4", "baft"}) for a producer that applies it itself, or None."""
        return None

    def fused_window_plan(self, x):
        """(geometry, input-quantiser spec, token permutation) when the LayerNorm in front of this attention can emit the
        quantiser's codes directly in window-major order (no padding: every 224-px stage), else None.  perm[t] = row of token t
        of an image after the cyclic shift + window partition (swin.py:103-131)."""
        if not (SWIN_ATTN_FUSE and x.is_cuda and x.dim() == 4):
            return None
        g = WindowGeometry(x, self.window_size, self.shift_size)
        if g.pad != (0, 0):
            return None
        spec = self._input_spec((g.B * g.nW, g.N, g.C))
        if spec is None:
            return None
        return g, spec, g._perm(x.device)[1]

    def window_forward_pre(self, g, pre):
        """The attention on window-major codes `pre` = (x_hat carrier, codes, geom) -> proj output, STILL window-major
        ((B*nW, N, C)): the caller's next LayerNorm pass reads it through the same permutation."""
        self.shift_size = g.ss
        add = g.addend(self.relative_position_bias_table, self.relative_position_index, self.num_heads)
        out = self._core(self, pre[0], (self.dim // self.num_heads) ** -0.5, add, pre_quant=pre)
        return self.proj(out)


class QAttention_swin(_SwinQBase):
    """swin_attention_and_mlp.py:65-251."""

    def __init__(self, m: ShiftedWindowAttention, weight_bits=8, input_bits=8, aq_learnable=True, wq_learnable=True,
                 weight_channelwise=True, input_channelwise=True, weight_quant_method="statsq", input_quant_method="lsq",
                 pretrained_initialized=False, **kwargs):
        self._base_init(m, weight_bits, input_bits, input_channelwise)
        kw = _qlinear_kwargs(weight_bits, input_bits, weight_channelwise, input_channelwise, weight_quant_method,
                             input_quant_method, aq_learnable, wq_learnable, pretrained_initialized)
        # `self.qkv` / `self.proj` (created by _base_init), not the source module's: swin_attention_and_mlp.py:89-90, :102-103
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

    _core = staticmethod(plain_attention_core)

    def _input_spec(self, win_shape):
        return self.qkv.fused_input_quant(win_shape)

    def forward(self, x):
        return self._window_forward(x, plain_attention_core)


class QAttention_swin_qkreparam(_SwinQBase):
    """swin_attention_and_mlp.py:253-461."""

    def __init__(self, m: ShiftedWindowAttention, weight_bits=8, input_bits=8, aq_learnable=True, wq_learnable=True,
                 symmetric=True, weight_channelwise=True, input_channelwise=True, weight_quant_method="statsq",
                 input_quant_method="lsq", pretrained_initialized=False, boundaryRange=0.005, **kwargs):
        self._base_init(m, weight_bits, input_bits, input_channelwise)
        C = m.qkv.in_features
        self.quant_x_4_qkv = LSQ_input(bit=input_bits, all_positive=(symmetric == False), learnable=aq_learnable,  # noqa: E712
                                       learanbaleBiasdim=C)
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
        del self.qkv

    def _make_qk_quant(self, wq_learnable, boundaryRange):
        return StatsQuantizer(num_bits=self.weight_bits, clip_learnable=wq_learnable)

    _core = staticmethod(qkr_attention_core)

    def _input_spec(self, win_shape):
        _, N, C = win_shape
        xin = self.quant_x_4_qkv
        if not (_ql.FUSE_NORM_QUANT and _attn_codes_ok(self, N, C)):
            return None
        return {"quant": xin.input_quant_fn, "b4": xin.move_b4.bias, "baft": xin.move_aft.bias}

    def forward(self, x):
        return self._window_forward(x, qkr_attention_core)


class QAttention_swin_qkreparam_4_cga(QAttention_swin_qkreparam):
    """swin_attention_and_mlp.py:463-675 (numerically identical; StatsQuantizer_specific_4_qkreparam_cga)."""

    def _make_qk_quant(self, wq_learnable, boundaryRange):
        return StatsQuantizer_specific_4_qkreparam_cga(num_bits=self.weight_bits, clip_learnable=wq_learnable,
                                                       boundaryRange=boundaryRange)
