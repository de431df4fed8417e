"""LSQ quantisers — drop-ins for src/quantization/quantizer/lsq.py (LsqQuantizer :515, LsqQuantizer4v :701,
LsqQuantizer4img :306, LsqQuantizer4Conv2d :384, LsqQuantizer4head_input :448, LsqQuantizerWeight :20).

All six share one fused HIP kernel pair (ofq_lsq_fwd / ofq_lsq_bwd); they differ in how the tensor maps onto
the kernel's [outer][S][inner] view and in M (the count in the 1/sqrt(hi*M) gradient scale).  The Q-modules
call `.quant(x, b4, baft, ...)` to fuse the LearnableBias pair (and QMLP's GELU) into the same kernel;
`.forward(x)` is the reference's plain signature.

The learnable step `s` is created lazily from the first batch exactly like the reference (init_from), so
`setup_alpha` must run before the optimizer is built (train.py:656-662)."""
import math

import torch
import torch.nn as nn

from ... import functional as F_ofq
from ... import ops


def _bounds(bit, all_positive):
    if all_positive:                                                   # lsq.py:519-526
        return (0, 1) if bit == 1 else (0, 2 ** bit - 1)
    return (-1, 1) if bit == 1 else (-(2 ** (bit - 1)), 2 ** (bit - 1) - 1)


class _LsqFn(torch.autograd.Function):
    """y = LSQ(pre(x) + b4) + baft with the closed-form backward."""

    @staticmethod
    def forward(ctx, x, s, b4, baft, geom, want_codes, need_values=True, pre_codes=None, link=None, fused=None):
        # link: dict shared with the single consumer of the codes (CodesLinearFn); when its backward fuses this
        # quantiser's backward into the dX GEMM epilogue it leaves the four gradients in link["done"]
        # fused: the spec this quantiser handed to the GEMM that produces x (fusable()); when that GEMM did not store x
        # (fused["producer"] present, x is a placeholder) the backward recomputes x from the producer's integer operands
        ctx.link = link
        ctx.sum_leaves = (s, b4, baft)
        ctx.fused = fused if (fused is not None and pre_codes is not None and "producer" in fused) else None
        if link is not None:
            link.update(x=x, s=s, b4=b4, geom=geom)
        if pre_codes is not None:
            # the producer GEMM's epilogue already applied this quantiser (ofq_qgemm_i8_nt_q): same codes, no second pass
            assert want_codes and not need_values
            y, codes = ops.placeholder((geom.outer * geom.S, geom.ldy), x.device), pre_codes
        else:
            y, codes = ops.lsq_fwd(x, s, b4, baft, geom, want_codes=want_codes, need_values=need_values)
        ctx.save_for_backward(x, s, b4)
        ctx.geom = geom
        ctx.has_bias = b4 is not None
        if codes is None:
            codes = torch.empty(0, dtype=torch.int8, device=x.device)
        ctx.mark_non_differentiable(codes)
        ctx.set_materialize_grads(False)          # no zero-filled int8 "gradient" for the codes output
        return y, codes

    @staticmethod
    def backward(ctx, gy, _gcodes):
        if ctx.link is not None and "done" in ctx.link:
            dx, ds, db4, dbaft = ctx.link.pop("done")
            ctx.link.clear()
            return dx, ds, db4, dbaft, None, None, None, None, None, None
        if ctx.link is not None:
            ctx.link.clear()
        if gy is None:
            return None, None, None, None, None, None, None, None, None, None
        x, s, b4 = ctx.saved_tensors
        g = ctx.geom
        gy = gy.contiguous()
        if ctx.fused is not None:
            q = ctx.fused
            n_out = q["producer"]["wcodes"].shape[0]
            with F_ofq.sum_scope(*ctx.sum_leaves):
                dx, ds, db4, dbaft = ops.qgemm_i8_lsq_bwd(gy.view(-1, n_out), q["producer"], q)
            return dx.view(x.shape), ds, db4, dbaft, None, None, None, None, None, None
        with F_ofq.sum_scope(*ctx.sum_leaves):
            dx, ds, db4, dbaft = ops.lsq_bwd(gy, x, s, b4, g)
        return dx.view(x.shape), ds, db4, dbaft, None, None, None, None, None, None


class _LsqBase(nn.Module):
    def __init__(self, bit, all_positive=False, per_channel=True, learnable=True, **kwargs):
        super().__init__()
        if bit == 1:
            raise ValueError("1-bit LSQ (sign) is not on the OFQ hot path")
        if not per_channel:
            raise ValueError("per_channel=False (one step per tensor) is not used by any OFQ recipe and is not implemented")
        self.bit = bit
        self.per_channel = per_channel
        self.all_positive = all_positive
        self.learnable = learnable
        self.thd_neg, self.thd_pos = _bounds(bit, all_positive)
        self.register_parameter("s", None)                              # lsq.py:541
        self.initialized_alpha = False

    # -- geometry: subclasses say how x maps onto [outer][S][inner] and what M is
    def _geom(self, x, bias_len, prologue, ldx, ldy):
        raise NotImplementedError

    def _init_value(self, xin):
        raise NotImplementedError

    def init_from(self, xin):
        init_val = self._init_value(xin.detach())
        self.s = nn.Parameter(init_val.to(xin.device).float().contiguous().clone(), requires_grad=bool(self.learnable))
        self.initialized_alpha = True

    def fusable(self, shape, b4, prologue):
        """Description of this quantiser for a producer GEMM epilogue (ops.qgemm_i8_nt fuse=...), or None when it cannot
        be applied there (not initialised yet, unsupported geometry).  `shape` is the shape quant() would be given."""
        if not self.initialized_alpha or self.s is None or b4 is None:
            return None
        geom = self._geom(tuple(shape), b4.numel(), prologue, None, None)
        if geom.inner % 16 or geom.bias_len % geom.inner:
            return None
        k = geom.bias_len // geom.inner                   # offset phases = quantiser rows per producer output row
        if geom.mode == 0:
            if k > 1 and (geom.inner % 128 or geom.S % k):
                return None
            extra = {"rowmul": k, "coldiv": geom.inner, "colmode": 0}
        elif geom.mode == 1 and k == 1:
            extra = {"rowmul": 1, "coldiv": geom.inner, "colmode": 1}
        else:
            return None
        spec = {"s": self.s.detach(), "S": geom.S if geom.mode == 0 else geom.inner, "gscale": geom.gscale,
                "b4": b4.detach(), "lo": geom.lo, "hi": geom.hi, "gelu": prologue == 1}
        spec.update(extra)
        return spec

    def quant(self, x, b4=None, baft=None, prologue=0, shape=None, ldx=None, ldy=None, out_shape=None,
              want_codes=False, need_values=True, pre_codes=None, link=None, fused=None, patch=None):
        """Fused (x [+gelu] + b4) -> LSQ -> + baft.  `shape` overrides x.shape for the geometry (used when x
        is a strided column slice).  patch = (width, ph, pw): values, codes (and the gradient the backward is given) in the im2col
        order of a stride == kernel convolution over x = (B, Cin, H, W) -- pass out_shape = (B * gh * gw, Cin * ph * pw)."""
        if not x.is_cuda:
            raise RuntimeError("ofq_amd LSQ: input must be on a HIP device; there is no CPU fallback")
        carrier = x.dim() > 0 and x.stride(-1) == 0          # zero-stride placeholder: the values only exist as codes
        if ldx is None and not carrier:
            x = x.contiguous()
        if carrier and (pre_codes is None or need_values):
            raise RuntimeError("ofq_amd LSQ: input is a code-only placeholder but its values are needed")
        shp = tuple(shape) if shape is not None else tuple(x.shape)
        if not self.initialized_alpha or self.s is None:
            xin = x.detach().reshape(shp) if ldx is None else x.detach()
            if prologue == 1:
                xin = torch.nn.functional.gelu(xin)
            if b4 is not None:
                xin = self._add_bias_for_init(xin, b4.detach())
            self.init_from(xin)
        geom = self._geom(shp, 0 if b4 is None else b4.numel(), prologue, ldx, ldy)
        if patch is not None:
            geom.patch = tuple(int(v) for v in patch)
        if pre_codes is not None and (need_values or not want_codes):
            pre_codes = None
        if link is not None and (need_values or not want_codes or geom.mode != 0 or geom.bias_len not in (0, geom.inner)
                                 or geom.inner <= 128 or ldx is not None or not torch.is_grad_enabled()):
            link = None                       # the fused backward needs codes-only output, per-token step, one offset phase
        if fused is not None and "producer" in fused and pre_codes is None:
            raise RuntimeError("ofq_amd LSQ: the producing GEMM kept no fp32 output, but its codes are not used")
        y, codes = _LsqFn.apply(x, self.s, b4, baft, geom, want_codes, need_values, pre_codes, link, fused)
        y = y.view(out_shape if out_shape is not None else shp)
        if want_codes:
            return y, codes, geom
        return y

    def _add_bias_for_init(self, xin, b4):
        return xin + b4

    def forward(self, x):
        return self.quant(x)

    def extra_repr(self):
        return "bit=%d, all_positive=%s, s_learnable=%s, per_channel=%s" % (self.bit, self.all_positive,
                                                                           self.learnable, self.per_channel)


class LsqQuantizer(_LsqBase):
    """Per-"token" step: s indexed by x.shape[-2] (lsq.py:556, :575).  x is (..., S, inner)."""

    def _geom(self, shp, bias_len, prologue, ldx, ldy):
        S, inner = shp[-2], shp[-1]
        n = 1
        for d in shp:
            n *= d
        outer = n // (S * inner)
        M = n // S                                                       # lsq.py:583-588
        return ops.LsqGeom(outer, S, inner, bias_len, 0, self.thd_neg, self.thd_pos, M, prologue, ldx, ldy)

    def _init_value(self, x):
        k = 4 if self.all_positive else 2                                # lsq.py:547-554
        a = x.abs().mean(dim=-1)
        while a.dim() > 1:
            a = a.mean(dim=0)
        return k * a / (self.thd_pos ** 0.5)

    def _add_bias_for_init(self, xin, b4):
        # bias over the flattened trailing axes: b4 has k*inner entries, row r uses block r % k
        inner = xin.shape[-1]
        k = b4.numel() // inner
        if k == 1:
            return xin + b4
        S = xin.shape[-2]
        return (xin.reshape(-1, S // k, k, inner) + b4.view(k, inner)).reshape(xin.shape)


class LsqQuantizer4v(_LsqBase):
    """Per-channel step: s indexed by the last dim (lsq.py:742, :761-766); M = product of leading dims."""

    def _geom(self, shp, bias_len, prologue, ldx, ldy):
        inner = shp[-1]
        n = 1
        for d in shp:
            n *= d
        rows = n // inner
        return ops.LsqGeom(rows, 1, inner, bias_len, 1, self.thd_neg, self.thd_pos, rows, prologue, ldx, ldy)

    def _init_value(self, x):
        k = 4 if self.all_positive else 2                                # lsq.py:732-737
        a = x.abs()
        while a.dim() > 1:
            a = a.mean(dim=0)
        return k * a / (self.thd_pos ** 0.5)

    def _add_bias_for_init(self, xin, b4):
        return xin + b4


class LsqQuantizer4img(_LsqBase):
    """8-bit image quantiser, step per input channel, signedness latched from the data (lsq.py:306-382)."""

    def __init__(self, bit=8, all_positive=False, per_channel=True, learnable=True, **kwargs):
        super().__init__(bit, all_positive, per_channel, learnable)
        self.register_buffer("signed", torch.zeros(1))                   # lsq.py:310
        self._signed_host = None     # host copy of `signed` (None: unknown, read the buffer once); see _latch

    def _load_from_state_dict(self, *a, **k):
        super()._load_from_state_dict(*a, **k)
        self._signed_host = None

    def sync_latch(self):
        """Forget the host copy of `signed` (call after writing the buffer from outside: broadcast, checkpoint load)."""
        self._signed_host = None

    def latched(self):
        if self._signed_host is None:
            self._signed_host = float(self.signed) != 0
        return self._signed_host

    def _latch(self, xin):
        """lsq.py:338-355: `signed` goes 0 -> 1 the first time a negative value is seen and never back.  The reference
        reads the buffer (a device sync) on every call; here the host keeps a copy, so a latched quantiser -- every
        normalised ImageNet batch -- costs no sync, and the data is only inspected while the quantiser is unsigned."""
        if not self.latched() and bool((xin.min() < -1e-5).item()):
            self.signed.data.fill_(1)
            self._signed_host = True

    def force_latch(self):
        """Another rank has seen a negative value (engine.GraphedTrainStep all-reduces the decision so that every rank
        re-captures together); DDP's buffer broadcast would hand over rank 0's flag on the next forward anyway."""
        self.signed.data.fill_(1)
        self._signed_host = True

    def _geom(self, shp, bias_len, prologue, ldx, ldy):
        B, Cc, Hh, Ww = shp
        return ops.LsqGeom(B, Cc, Hh * Ww, bias_len, 0, self.thd_neg, self.thd_pos, B * Hh * Ww, prologue)

    def _init_value(self, x):
        k = 4 if self.all_positive else 2                                # lsq.py:322-323
        return k * x.abs().mean(dim=-1).mean(dim=-1).mean(dim=0) / (self.thd_pos ** 0.5)

    def _add_bias_for_init(self, xin, b4):
        return xin + b4.view(xin.shape[-1], xin.shape[-2])               # qbias.py:21

    def latch_input(self, x, b4):
        return x.detach() if b4 is None else x.detach() + b4.detach().view(x.shape[-1], x.shape[-2])

    def quant(self, x, b4=None, baft=None, **kw):
        # inside a stream capture the host cannot look at the data: engine.GraphedTrainStep takes the decision before the
        # capture / replay and re-captures when it flips (the clamp bounds are arguments of the captured launches)
        if not self.latched() and not torch.cuda.is_current_stream_capturing():
            self._latch(self.latch_input(x, b4))
        self.thd_neg, self.thd_pos = _bounds(self.bit, not self.latched())
        return super().quant(x, b4, baft, **kw)


class LsqQuantizer4Conv2d(_LsqBase):
    """Conv weight (O,I,kh,kw), step per out-channel, M = I*kh*kw (lsq.py:384-446)."""

    def __init__(self, bit=8, all_positive=False, per_channel=True, learnable=True, **kwargs):
        super().__init__(bit, False, per_channel, learnable)
        self.all_positive = all_positive

    def _geom(self, shp, bias_len, prologue, ldx, ldy):
        O = shp[0]
        inner = 1
        for d in shp[1:]:
            inner *= d
        return ops.LsqGeom(1, O, inner, 0, 0, self.thd_neg, self.thd_pos, inner)

    def _init_value(self, x):
        k = 4 if self.all_positive else 2                                # lsq.py:405-406
        return k * x.abs().mean(dim=-1).mean(dim=-1).mean(dim=-1) / (self.thd_pos ** 0.5)


class LsqQuantizerWeight(_LsqBase):
    """2-D weight, step per row, M = in_features (lsq.py:20-109)."""

    def _geom(self, shp, bias_len, prologue, ldx, ldy):
        return ops.LsqGeom(1, shp[0], shp[1], 0, 0, self.thd_neg, self.thd_pos, shp[1])

    def _init_value(self, x):
        return 2 * x.abs().mean(dim=-1) / (self.thd_pos ** 0.5)          # lsq.py:54


class LsqQuantizer4head_input(_LsqBase):
    """Per-tensor scalar step, M = numel (lsq.py:448-513)."""

    def _geom(self, shp, bias_len, prologue, ldx, ldy):
        n = 1
        for d in shp:
            n *= d
        inner = shp[-1]
        return ops.LsqGeom(n // inner, 1, inner, bias_len, 0, self.thd_neg, self.thd_pos, n, prologue)

    def _init_value(self, x):
        return (x.abs().mean() * 2 / (self.thd_pos ** 0.5)).reshape(1)   # lsq.py:480

    def _add_bias_for_init(self, xin, b4):
        return xin + b4
