"""QLinear / QMLP / LSQ_input / LSQ_QConv2d / LSQ_QLinear4head — drop-ins for
src/quantization/modules/qlinear.py (:28, :89, :12, :138, :193): same class names, constructor
signatures, parameter names and shapes (= checkpoint format), same error behaviour.
Every forward is a short chain of HIP kernels: StatsQ / LSQ-weight -> fused offset+LSQ+offset -> MFMA GEMM."""
import os

import torch
import torch.nn as nn

from .qbias import LearnableBias, LearnableBias4img
from ..quantizer.statsq import StatsQuantizer
from ..quantizer.lsq import (LsqQuantizer, LsqQuantizerWeight, LsqQuantizer4img, LsqQuantizer4Conv2d,
                             LsqQuantizer4head_input)
from ...deit_vision_transformer import Mlp, to_2tuple
from ... import ops
from ...functional import (LinearFn, codes_linear, codes_linear_ok, codes_only_ok, CodeWeightLinearFn,
                           code_weight_linear_ok)


# Exact integer-code GEMMs (int8 forward, bf16-split dX) instead of the fp32-MFMA GEMM on fake-quant values.
# Same mathematical function; toggled off by the parity tests that compare the two paths.
USE_CODE_GEMM = True
# The plain (non-QKR) attention core on integer codes as well (int8 scores / P.V, bf16-split backward) instead of the
# fp32-MFMA batched GEMMs on fake-quant values (A/B switch: OFQ_NO_PLAIN_ATTN_CODES=1)
PLAIN_ATTN_CODES = os.environ.get("OFQ_NO_PLAIN_ATTN_CODES") is None
# fc1's GEMM epilogue emits fc2's input codes (A/B switch for bench runs: OFQ_NO_EPILOGUE_FUSE=1)
FUSE_NEXT_CODES = os.environ.get("OFQ_NO_EPILOGUE_FUSE") is None
FUSE_NEXT_CODES_MLP = os.environ.get("OFQ_NO_EPILOGUE_FUSE_MLP") is None
# The input quantiser's backward can run in the epilogue of the layer's dX GEMM (ofq_qgemm_bf16s_nt_lsq: dX never
# goes to HBM).  Correct and tested, but OFF by default: measured on MI355X it loses -- the 2-waves-per-SIMD GEMM kernel
# executes the division-heavy LSQ arithmetic at a fraction of the rate of the 5-waves-per-SIMD elementwise kernel
# (+75 us per GEMM launch vs 51 us for the separate kernel; round 6, two-plane form, both sites of a block: 21.03 -> 22.05 ms per
# step).  OFQ_LSQ_BWD_FUSE=1 turns it on.
FUSE_LSQ_BWD = os.environ.get("OFQ_LSQ_BWD_FUSE") is not None
# the stem's image quantiser writes / reads the convolution's im2col order itself (A/B switch: OFQ_NO_STEM_PATCH_LAYOUT=1)
STEM_PATCH_LAYOUT = os.environ.get("OFQ_NO_STEM_PATCH_LAYOUT") is None
# LayerNorm + the per-token LSQ of its single consumer in one kernel each way (A/B switch: OFQ_NO_NORM_QUANT_FUSE=1)
FUSE_NORM_QUANT = os.environ.get("OFQ_NO_NORM_QUANT_FUSE") is None
# A GEMM whose epilogue applies its only consumer's quantiser writes that quantiser's codes and nothing else; the
# quantiser's backward recomputes the fp32 layer output from the integer operands (ofq_qgemm_i8_lsq_bwd) instead of
# reading a saved copy: -8 B/element of HBM traffic.  Sites: "qkx" (H*C wide, the largest activation of a QKR block),
# "fc1" (GELU in front of fc2's quantiser), "v".  Measured on MI355X the recompute kernel wins for qkx only (its
# epilogue is the consumer's whole backward at two waves per SIMD; the GELU variant loses to the HBM-bound elementwise
# kernel), so that is the default.  OFQ_RECOMPUTE=qkx,fc1,v / OFQ_RECOMPUTE= (none) select others.
RECOMPUTE_SITES = frozenset(x for x in os.environ.get("OFQ_RECOMPUTE", "qkx").split(",") if x)


class LSQ_input(nn.Module):
    """qlinear.py:12-26 — offset -> LSQ -> offset, shared by the V and QK branches of the QKR attention."""

    def __init__(self, bit=2, all_positive=False, learnable=True, learanbaleBiasdim=192):
        super().__init__()
        self.input_bits = bit
        self.all_positive = all_positive
        self.learnable = learnable
        self.input_quant_fn = LsqQuantizer(bit=bit, all_positive=all_positive, learnable=learnable)
        self.move_b4 = LearnableBias(learanbaleBiasdim)
        self.move_aft = LearnableBias(learanbaleBiasdim)

    def forward(self, input, want_codes=False, need_values=True):
        return self.input_quant_fn.quant(input, self.move_b4.bias, self.move_aft.bias, want_codes=want_codes,
                                         need_values=need_values)


class QLinear(nn.Linear):
    def __init__(self, *kargs, m: torch.nn.Linear, weight_bits=8, input_bits=8, aq_learnable=True, wq_learnable=True,
                 symmetric=True, weight_channelwise=True, input_channelwise=True, weight_quant_method="statsq",
                 input_quant_method="lsq", pretrained_initialized=False, **kwargs):
        super().__init__(m.in_features, m.out_features, bias=True)              # always a bias (qlinear.py:34)
        self.weight_bits = weight_bits
        self.input_bits = input_bits
        self.aq_learnable = aq_learnable
        self.wq_learnable = wq_learnable
        self.symmetric = symmetric
        self.weight_channelwise = weight_channelwise
        self.input_channelwise = input_channelwise
        self.weight_quant_method = weight_quant_method
        self.input_quant_method = input_quant_method
        self.input_quant_fn = LsqQuantizer(bit=input_bits, all_positive=(symmetric == False), learnable=aq_learnable)  # noqa: E712
        self.pretrained_initialized = pretrained_initialized
        if pretrained_initialized != False:  # noqa: E712
            self.weight = torch.nn.Parameter(m.weight.detach())
            if m.bias is not None:
                self.bias = torch.nn.Parameter(m.bias.detach())
        if weight_quant_method == "statsq":
            self.statsq_fn = StatsQuantizer(num_bits=self.weight_bits, clip_learnable=wq_learnable).to(m.weight.device)
        else:
            raise ValueError("Unknown quant_method")
        self.move_b4 = LearnableBias(self.weight.shape[1])
        self.move_aft = LearnableBias(self.weight.shape[1])
        self._prologue = 0          # 1: exact GELU fused in front of the input quantiser (set by QMLP for fc2)

    def code_path(self):
        return USE_CODE_GEMM and codes_linear_ok(self.in_features, self.statsq_fn, self.input_quant_fn)

    def input_fuse_spec(self, in_shape):
        """What a producer's GEMM epilogue needs to emit this layer's input codes itself (None: not possible)."""
        if not (self.code_path() and codes_only_ok(self.in_features, self.out_features)):
            return None
        return self.input_quant_fn.fusable(in_shape, self.move_b4.bias, self._prologue)

    def fused_input_quant(self, in_shape):
        """The input quantiser, for a producer that can apply it itself (LayerNorm + LSQ in one kernel); None if this
        layer needs the fp32 values or is not on the code path."""
        if not (FUSE_NORM_QUANT and self.code_path() and codes_only_ok(self.in_features, self.out_features)
                and self._prologue == 0):
            return None
        return {"quant": self.input_quant_fn, "b4": self.move_b4.bias, "baft": self.move_aft.bias}

    def forward(self, input, fuse_next=None, pre_codes=None, pre_quant=None, fused=None):
        """fuse_next: input_fuse_spec() of the layer consuming this output (its codes come back in fuse_next["codes_out"]);
        pre_codes: this layer's own input codes when a producer already computed them;
        pre_quant: (x_hat carrier, codes, geom) when the producer ran this layer's whole input quantiser (norm_quant)."""
        if self.weight_quant_method != "statsq":
            raise ValueError("Unknown quant_method")
        if pre_quant is not None:
            xq, xcodes, geom = pre_quant
            return codes_linear(xq, xcodes, geom, self.input_quant_fn, self.move_aft.bias, self.weight,
                                self.statsq_fn, self.bias, fuse=fuse_next)
        if self.code_path():
            # this quantiser has exactly one consumer (the GEMM below): its backward can ride in the dX GEMM's epilogue
            link = {} if FUSE_LSQ_BWD else None
            xq, xcodes, geom = self.input_quant_fn.quant(input, self.move_b4.bias, self.move_aft.bias,
                                                         prologue=self._prologue, want_codes=True,
                                                         need_values=not codes_only_ok(self.in_features, self.out_features),
                                                         pre_codes=pre_codes, link=link, fused=fused)
            return codes_linear(xq, xcodes, geom, self.input_quant_fn, self.move_aft.bias, self.weight,
                                self.statsq_fn, self.bias, fuse=fuse_next, lsq_link=link)
        weight = self.statsq_fn(self.weight)                                     # qlinear.py:62
        xq = self.input_quant_fn.quant(input, self.move_b4.bias, self.move_aft.bias, prologue=self._prologue)
        return LinearFn.apply(xq, weight, self.bias)                             # qlinear.py:69-71

    def extra_repr(self):
        return (f"act_bit={self.input_bits}, weight_bit={self.weight_bits}, act_all_positive={not self.symmetric}, "
                f"wq_learnable={self.wq_learnable}, aq_learnable={self.aq_learnable}, "
                f"weight_channelwise ={self.weight_channelwise}, input_channelwise ={self.input_channelwise}, "
                f"weight_quant_method={self.weight_quant_method}, activation_quant_method={self.input_quant_method}, "
                f"pretrained_initialized = {self.pretrained_initialized}")


class QMLP(Mlp):
    def __init__(self, *kargs, m: Mlp, weight_bits=8, input_bits=8, aq_learnable=True, wq_learnable=True,
                 weight_channelwise=True, input_channelwise=True, weight_quant_method="statsq",
                 input_quant_method="lsq", act_layer=nn.GELU, pretrained_initialized=False, **kwargs):
        super().__init__(in_features=m.in_features, hidden_features=m.hidden_features, out_features=m.out_features,
                         drop=m.drop)
        drop_probs = to_2tuple(self.drop)
        common = dict(weight_bits=weight_bits, input_bits=input_bits, aq_learnable=aq_learnable,
                      wq_learnable=wq_learnable, weight_channelwise=weight_channelwise,
                      input_channelwise=input_channelwise, weight_quant_method=weight_quant_method,
                      input_quant_method=input_quant_method, pretrained_initialized=pretrained_initialized)
        self.fc1 = QLinear(m=m.fc1, symmetric=True, **common)
        self.act_layer = act_layer
        if act_layer == "rprelu":
            raise ValueError("rprelu is not part of any shipped OFQ recipe")
        self.act = act_layer() if act_layer != "None" else nn.Identity()
        self.drop1 = nn.Dropout(drop_probs[0])
        self.fc2 = QLinear(m=m.fc2, symmetric=False, **common)                   # unsigned input (qlinear.py:118-120)
        self.drop2 = nn.Dropout(drop_probs[1])
        # exact GELU followed by dropout(p=0) is folded into fc2's input-quantiser kernel
        self._fuse_gelu = isinstance(self.act, nn.GELU) and getattr(self.act, "approximate", "none") == "none" \
            and drop_probs[0] == 0
        self.fc2._prologue = 1 if self._fuse_gelu else 0

    def fused_input_quant(self, in_shape):
        return self.fc1.fused_input_quant(in_shape)

    def forward(self, x, pre_quant=None):
        if FUSE_NEXT_CODES and FUSE_NEXT_CODES_MLP and self._fuse_gelu and self.fc1.code_path():
            # fc1's GEMM epilogue also applies GELU + fc2's offset and LSQ, so fc2 never re-reads the fp32 activation
            spec = self.fc2.input_fuse_spec(tuple(x.shape[:-1]) + (self.fc1.out_features,))
            if spec is not None and "fc1" in RECOMPUTE_SITES and not FUSE_LSQ_BWD:
                spec["store_y"] = False
            h = self.fc1(x, fuse_next=spec, pre_quant=pre_quant)
            x = self.fc2(h, pre_codes=None if spec is None else spec.get("codes_out"), fused=spec)
            return self.drop2(x)
        x = self.fc1(x, pre_quant=pre_quant)
        if not self._fuse_gelu:
            x = self.drop1(self.act(x))
        x = self.fc2(x)
        return self.drop2(x)


class LSQ_QConv2d(nn.Conv2d):
    """W8A8 patch embedding (qlinear.py:138-191).  The stride==kernel conv is an im2col view + MFMA GEMM."""

    def __init__(self, *kargs, m: torch.nn.Conv2d, weight_bits=8, input_bits=8, aq_learnable=True, wq_learnable=True,
                 symmetric=True, weight_channelwise=True, input_channelwise=True, weight_quant_method="lsq",
                 input_quant_method="lsq", pretrained_initialized=False, **kwargs):
        super().__init__(in_channels=m.in_channels, out_channels=m.out_channels, kernel_size=m.kernel_size,
                         stride=m.stride, padding=m.padding, dilation=m.dilation, groups=m.groups, bias=True)
        self.weight_bits = weight_bits
        self.input_bits = input_bits
        self.aq_learnable = aq_learnable
        self.wq_learnable = wq_learnable
        self.symmetric = symmetric
        self.weight_channelwise = weight_channelwise
        self.input_channelwise = input_channelwise
        self.weight_quant_method = weight_quant_method
        self.input_quant_method = input_quant_method
        self.input_quant_fn = LsqQuantizer4img(bit=input_bits, all_positive=(symmetric == False), learnable=aq_learnable)  # noqa: E712
        self.pretrained_initialized = pretrained_initialized
        if pretrained_initialized != False:  # noqa: E712
            self.weight = torch.nn.Parameter(m.weight.detach())
            if m.bias is not None:
                self.bias = torch.nn.Parameter(m.bias.detach())
        self.lsqw_fn = LsqQuantizer4Conv2d(bit=self.weight_bits, learnable=aq_learnable).to(m.weight.device)
        self.move_b4 = LearnableBias4img(224 * 224)                              # qlinear.py:163-164
        self.move_aft = LearnableBias4img(224 * 224)
        if tuple(self.stride) != tuple(self.kernel_size) or tuple(self.padding) != (0, 0) or self.groups != 1:
            raise ValueError("LSQ_QConv2d: only the non-overlapping patch-embedding conv is on the hot path")

    def forward(self, input):
        K = self.weight[0].numel()
        code_dx = (USE_CODE_GEMM and input.is_cuda and torch.is_grad_enabled() and self.lsqw_fn.initialized_alpha
                   and code_weight_linear_ok(self.out_channels, K, self.lsqw_fn))
        if code_dx:
            # W_hat = step[o] * code[o, k]: the codes (int8) feed the two-plane code GEMM of the input gradient
            weight, wcodes, wgeom = self.lsqw_fn.quant(self.weight, want_codes=True)      # qlinear.py:168
        else:
            weight = self.lsqw_fn(self.weight)                                   # qlinear.py:168
        xin = self.input_quant_fn
        if xin.initialized_alpha and not xin.latched() and not torch.cuda.is_current_stream_capturing():
            # the signedness latch (lsq.py:338-355) is taken HERE, before the choice below reads it: quant() would take it a
            # moment later, and the step in which it flips would pick another weight-gradient kernel than the captured step
            # that engine.GraphedTrainStep re-captures for the same batch
            xin._latch(xin.latch_input(input, self.move_b4.bias))
        # the weight gradient on the image quantiser's codes as well: int8 codes (a signed, latched quantiser), geometry the
        # wide dW kernel takes
        code_dw = (code_dx and xin.initialized_alpha and xin.s is not None and xin.latched() and xin.bit <= 8
                   and K % 384 == 0 and self.out_channels % 4 == 0 and self.weight.requires_grad)
        xcodes = None
        B, Cin, Hh, Ww = input.shape
        kh, kw = self.kernel_size
        gh, gw = Hh // kh, Ww // kw
        # im2col of a stride==kernel conv is a pure permutation, (B, gh*gw, Cin*kh*kw): the image quantiser writes its values and
        # codes in that order itself, and its backward reads the GEMM's input gradient in it (round 6, ofq_lsq_fwd_patch /
        # _bwd_patch: three permute copies per step gone); STEM_PATCH_LAYOUT False = the copies
        patch = (Ww, kh, kw) if (STEM_PATCH_LAYOUT and input.is_cuda and input.dtype == torch.float32 and Ww % 4 == 0 and kw % 4 == 0
                                 and Hh % kh == 0 and Ww % kw == 0 and xin.initialized_alpha and xin.s is not None) else None
        pkw = dict(patch=patch, out_shape=(B * gh * gw, Cin * kh * kw)) if patch is not None else {}
        if code_dw:
            xq, xcodes, xgeom = xin.quant(input, self.move_b4.bias, self.move_aft.bias, want_codes=True, **pkw)   # :171-173
        else:
            xq = xin.quant(input, self.move_b4.bias, self.move_aft.bias, **pkw)   # :171-173
        if patch is not None:
            cols = xq
        else:
            cols = xq.view(B, Cin, gh, kh, gw, kw).permute(0, 2, 4, 1, 3, 5).reshape(B * gh * gw, Cin * kh * kw)
        if code_dx:
            steps = ops.lsq_eff_scale_vec(self.lsqw_fn.s.detach(), wgeom.gscale)  # the step VALUE the fake-quant weights carry
            xaux = None
            if xcodes is not None and xin.thd_neg >= -128 and xin.thd_pos <= 127:
                if patch is not None:
                    qx = xcodes.view(B * gh * gw, Cin * kh * kw)
                else:
                    qx = xcodes.view(B, Cin, gh, kh, gw, kw).permute(0, 2, 4, 1, 3, 5).reshape(B * gh * gw, Cin * kh * kw)
                ax = ops.lsq_eff_scale_vec(xin.s.detach(), xgeom.gscale, kh * kw)                             # step of column k
                boff = self.move_aft.bias.detach().view(gh, kh, gw, kw).permute(0, 2, 1, 3).reshape(gh * gw, kh * kw).repeat(1, Cin)
                if getattr(self, "_ones32", None) is None or self._ones32.device != input.device:
                    self._ones32 = torch.ones(32, dtype=torch.float32, device=input.device)
                xaux = {"qx": qx, "ax": ax, "boff": boff.contiguous(), "ones": self._ones32}
            out = CodeWeightLinearFn.apply(cols, weight.view(self.out_channels, -1), self.bias, wcodes.view(self.out_channels, K),
                                           steps, xaux)                          # :174
        else:
            out = LinearFn.apply(cols, weight.view(self.out_channels, -1), self.bias)      # :174
        return out.view(B, gh, gw, self.out_channels).permute(0, 3, 1, 2)

    def extra_repr(self):
        return (f"act_bit={self.input_bits}, weight_bit={self.weight_bits}, act_all_positive={not self.symmetric}, "
                f"wq_learnable={self.wq_learnable}, aq_learnable={self.aq_learnable}, "
                f"weight_quant_method={self.weight_quant_method}, activation_quant_method={self.input_quant_method}, "
                f"pretrained_initialized = {self.pretrained_initialized}")


QConv2d = LSQ_QConv2d   # north_star's name; the reference only has a commented import of it (modules/utils.py:4)


class LSQ_QLinear4head(nn.Linear):
    """W8A8 classifier heads (qlinear.py:193-252)."""

    def __init__(self, *kargs, m: torch.nn.Linear, weight_bits=8, input_bits=8, aq_learnable=True, wq_learnable=True,
                 symmetric=True, weight_channelwise=True, input_channelwise=True, weight_quant_method="statsq",
                 input_quant_method="lsq", pretrained_initialized=False, **kwargs):
        super().__init__(m.in_features, m.out_features, bias=True)
        self.weight_bits = weight_bits
        self.input_bits = input_bits
        self.aq_learnable = aq_learnable
        self.wq_learnable = wq_learnable
        self.symmetric = symmetric
        self.weight_channelwise = weight_channelwise
        self.input_channelwise = input_channelwise
        self.weight_quant_method = weight_quant_method
        self.input_quant_method = input_quant_method
        self.input_quant_fn = LsqQuantizer4head_input(bit=input_bits, all_positive=(symmetric == False), learnable=aq_learnable)  # noqa: E712
        self.pretrained_initialized = pretrained_initialized
        if pretrained_initialized != False:  # noqa: E712
            self.weight = torch.nn.Parameter(m.weight.detach())
            if m.bias is not None:
                self.bias = torch.nn.Parameter(m.bias.detach())
        if weight_quant_method == "lsq":
            self.lsqw_fn = LsqQuantizerWeight(bit=self.weight_bits, per_channel=weight_channelwise,
                                              learnable=wq_learnable).to(m.weight.device)
        else:
            raise ValueError("Unknown quant_method")
        self.move_b4 = LearnableBias(self.weight.shape[1])
        self.move_aft = LearnableBias(self.weight.shape[1])

    def forward(self, input):
        if self.weight_quant_method == "lsq":
            weight = self.lsqw_fn(self.weight)                                   # qlinear.py:227
        else:
            raise ValueError("Unknown quant_method")
        xq = self.input_quant_fn.quant(input, self.move_b4.bias, self.move_aft.bias)
        return LinearFn.apply(xq, weight, self.bias)
