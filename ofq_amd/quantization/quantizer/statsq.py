"""StatsQuantizer — drop-in for src/quantization/quantizer/statsq.py:122-150 (and its CGA twin :154-193,
which is numerically identical in value and gradient).  Forward is one HIP kernel
(ofq_statsq_fwd); backward is the straight-through identity, so no kernel runs."""
import torch
import torch.nn as nn

from ... import ops


class _StatsQFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, weight, bits, holder, want_codes):
        out, s, codes = ops.statsq_fwd(weight, bits, want_levels=want_codes, odd_codes=want_codes)
        holder._s_dev = s
        holder._codes = codes                      # int8 (2L+1), [out][in]: operand of the exact i8 forward GEMM
        holder._codesT = None                      # bf16 [in][out], built on demand for the backward GEMM
        return out

    @staticmethod
    def backward(ctx, g):
        return g, None, None, None    # statsq.py:148: Wq.detach() - W.detach() + W  => dW = g


class StatsQuantizer(nn.Module):
    def __init__(self, num_bits, clip_learnable):
        super().__init__()
        self.num_bits = num_bits
        self.clip_val = nn.Parameter(torch.Tensor([2.0]), requires_grad=False)     # statsq.py:126-128
        self._s_dev = None
        self._codes = None
        self._codesT = None

    def codes_T(self):
        """Weight codes transposed to [in][out] as bf16 (exact small integers) for dX = dY @ W_hat."""
        if self._codesT is None:
            self._codesT = ops.codes_transpose_bf16(self._codes)
        return self._codesT

    @property
    def s(self):
        """Per-row scale of the last forward as a CPU tensor (the reference copies it to the host on every
        call, statsq.py:143; here the copy happens only when somebody reads it)."""
        return None if self._s_dev is None else self._s_dev.detach().cpu()

    def forward(self, weight, want_codes=False):
        if weight.dim() != 2:
            raise ValueError("StatsQuantizer: only 2-D weights are on the hot path (statsq.py:137-138)")
        return _StatsQFn.apply(weight, self.num_bits, self, bool(want_codes) and self.num_bits <= 7)

    def extra_repr(self):
        return "num_bits=%d" % self.num_bits


class StatsQuantizer_specific_4_qkreparam_cga(StatsQuantizer):
    """statsq.py:154-193.  Its training-time boundary loop only touches a detached branch, so values and
    gradients equal StatsQuantizer bit for bit (pinned by tests/golden g1 'G10')."""

    def __init__(self, num_bits, clip_learnable, boundaryRange=0.005):
        super().__init__(num_bits, clip_learnable)
        self.boundaryRange = boundaryRange
