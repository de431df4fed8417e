"""StatsQuantizer — drop-in for src/quantization/quantizer/statsq.py:122-150 (and its CGA twin :154-193,
which is numerically identical in value and gradient).  Forward is one HIP kernel
(ofq_statsq_fwd); backward is the straight-through identity, so no kernel runs."""
import torch
import torch.nn as nn

from ... import ops


class _StatsQFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, weight, bits, holder, want_codes, rvec=None, need_values=True, want_T=False):
        holder._r = None
        if holder._scale_override is not None:
            # test hook (tests/test_depth12_gpu.py): quantise with a GIVEN per-row scale (ofq_statsq_fwd, scale_given = 1)
            # instead of 2 * mean|W|, e.g. the torch-CPU value of the same weights, which can differ from the kernel's
            # correctly rounded fp64 row sum by one ulp.  The unfused operand path: codes^T and the offset row-dots are
            # built on demand by their consumers.
            out, s, codes = ops.statsq_fwd(weight, bits, want_levels=want_codes, scale=holder._scale_override.to(weight.device),
                                           odd_codes=want_codes)
            holder._s_dev, holder._codes, holder._codesT = s, codes, None
            return out
        if want_codes and (rvec is not None or not need_values):
            # code path of a linear layer: scale, codes, transposed bf16 codes and the offset row-dot in ONE launch;
            # the fp32 fake-quant values are not written (nothing reads them), a zero-stride tensor carries the edge
            if not need_values:
                # what refresh_weight_codes() / all_wqk() recompute in bulk.  A non-leaf weight (W_qk) is NOT kept: the
                # reference would pin its grad_fn -- and through it the AccumulateGrad nodes of q / k -- across steps
                holder._last_args = (weight if weight.is_leaf else None, bits, rvec, want_T)
                pre = holder._pre
                if (pre is not None and pre[0].data_ptr() == weight.data_ptr() and pre[0].shape == weight.shape
                        and (pre[1] is None) == (rvec is None) and (rvec is None or pre[1].data_ptr() == rvec.data_ptr())
                        and (pre[2] or not want_T)):
                    holder._s_dev, holder._codes, holder._codesT, holder._r = pre[3]
                    return ops.placeholder(tuple(weight.shape), weight.device)
            out, s, codes, codesT, r = ops.statsq_codes_fwd(weight, bits, rvec=rvec, need_values=need_values,
                                                            want_T=want_T)
            holder._s_dev, holder._codes, holder._codesT, holder._r = s, codes, codesT, r
            return out
        out, s, codes = ops.statsq_fwd(weight, bits, want_levels=want_codes, odd_codes=want_codes)
        holder._s_dev = s
        holder._codes = codes                      # int8 (2L+1), [out][in]: operand of the exact i8 forward GEMM
        holder._codesT = None                      # bf16 [in][out], built on demand for the backward GEMM
        return out

    @staticmethod
    def backward(ctx, g):
        return g, None, None, None, None, None, None    # statsq.py:148: Wq.detach() - W.detach() + W  => dW = g


class StatsQuantizer(nn.Module):
    def __init__(self, num_bits, clip_learnable):
        super().__init__()
        self.num_bits = num_bits
        self.clip_val = nn.Parameter(torch.Tensor([2.0]), requires_grad=False)     # statsq.py:126-128
        self._s_dev = None
        self._codes = None
        self._codesT = None
        self._r = None
        self._last_args = None      # (leaf weight or None, bits, rvec, want_T) of the last code-path forward
        self._pre = None            # (weight, rvec, has_T, (s, codes, codesT, r)) filled by engine.refresh_weight_codes()
        self._scale_override = None  # test hook: a per-row scale to use instead of 2 * mean|W| (see _StatsQFn.forward)

    def codes_T(self):
        """Weight codes transposed to [in][out] as bf16 (exact small integers) for dX = dY @ W_hat."""
        if self._codesT is None:
            self._codesT = ops.codes_transpose_16(self._codes)      # fp16 (two-plane backward) or bf16, ops.GRAD_PLANES
        return self._codesT

    @property
    def s(self):
        """Per-row scale of the last forward as a CPU tensor (the reference copies it to the host on every
        call, statsq.py:143; here the copy happens only when somebody reads it)."""
        return None if self._s_dev is None else self._s_dev.detach().cpu()

    def forward(self, weight, want_codes=False, rvec=None, need_values=True):
        """rvec / need_values: only meaningful with want_codes (see _StatsQFn.forward)."""
        if weight.dim() != 2:
            raise ValueError("StatsQuantizer: only 2-D weights are on the hot path (statsq.py:137-138)")
        wc = bool(want_codes) and self.num_bits <= 7
        # (grad mode is off inside Function.forward, so whether the backward operand is wanted is decided here)
        return _StatsQFn.apply(weight, self.num_bits, self, wc, rvec if wc else None, need_values or not wc,
                               torch.is_grad_enabled())

    def extra_repr(self):
        return "num_bits=%d" % self.num_bits


class StatsQuantizer_specific_4_qkreparam_cga(StatsQuantizer):
    """statsq.py:154-193.  Its training-time boundary loop only touches a detached branch, so values and
    gradients equal StatsQuantizer bit for bit (pinned by tests/golden g1 'G10')."""

    def __init__(self, num_bits, clip_learnable, boundaryRange=0.005):
        super().__init__(num_bits, clip_learnable)
        self.boundaryRange = boundaryRange
