"""AdamW on the HIP multi-tensor kernel (csrc/adamw.hip) -- SURVEY.md 8(f) rank 1.  Same update rule, hyper-parameters,
param-group handling and state-dict layout (`step`, `exp_avg`, `exp_avg_sq`) as torch.optim.AdamW (which the reference
gets from timm's create_optimizer_v2, train.py:662), one launch per parameter group instead of a chain of foreach
kernels, and the CGA weight freeze (cga.py:953-1013) folded into the same pass via `set_frozen`."""
import numpy as np
import torch

from . import ops


class FusedAdamW(torch.optim.Optimizer):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2):
        if lr < 0 or eps < 0 or not 0 <= betas[0] < 1 or not 0 <= betas[1] < 1 or weight_decay < 0:
            raise ValueError("FusedAdamW: invalid hyper-parameter")
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay))
        self._frozen = {}            # id(param) -> {0,1} fp32 mask of the parameter's shape (CGA), or absent
        self._shared_step = {}       # group index -> the cpu `step` tensor shared by the group's parameters

    def set_frozen(self, param, mask):
        """CGA: elements with mask != 0 take no update this step (gradient masked, weight restored)."""
        if mask is None:
            self._frozen.pop(id(param), None)
        else:
            self._frozen[id(param)] = mask

    def clear_frozen(self):
        self._frozen.clear()

    def _table(self, plist):
        """Host array of {p, g, m, v, frozen, n} (the kernel launch copies it into its arguments: nothing goes to the
        device ahead of the launch, so fresh gradient tensors every step cost nothing)."""
        tens = np.empty((len(plist), 6), dtype=np.int64)
        for i, p in enumerate(plist):
            st = self.state[p]
            f = self._frozen.get(id(p))
            tens[i] = (p.data_ptr(), p.grad.data_ptr(), st["exp_avg"].data_ptr(), st["exp_avg_sq"].data_ptr(),
                       0 if f is None else f.data_ptr(), p.numel())
        return tens

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        lib = ops.lib()
        assert lib.ofq_adamw_tensor_entry_bytes() == 48
        for gi, group in enumerate(self.param_groups):
            plist = [p for p in group["params"] if p.grad is not None]
            if not plist:
                continue
            for p in plist:
                if not p.is_cuda or p.dtype != torch.float32 or not p.is_contiguous() or not p.grad.is_contiguous():
                    raise RuntimeError("FusedAdamW: contiguous fp32 parameters on a HIP device only (no CPU fallback)")
                st = self.state[p]
                if not st:
                    st["step"] = torch.tensor(0.0)
                    st["exp_avg"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                    st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.preserve_format)
            # all tensors of a group take the same number of steps (as in torch: one `step` per tensor, advanced together);
            # they share ONE cpu tensor object, advanced once, instead of 150 fresh tensors per step
            first = self.state[plist[0]]["step"]
            t = int(first.item() if torch.is_tensor(first) else first) + 1
            shared = self._shared_step.get(gi)
            if shared is None or any(self.state[p]["step"] is not shared for p in plist):
                shared = self._shared_step[gi] = torch.tensor(float(t))
                for p in plist:
                    self.state[p]["step"] = shared
            else:
                shared.fill_(float(t))
            b1, b2 = group["betas"]
            tens = self._table(plist)
            ops._chk(lib.ofq_adamw_multi(tens.ctypes.data, len(plist), float(group["lr"]), float(b1), float(b2),
                                         float(group["eps"]), float(group["weight_decay"]), 1.0 - b1 ** t, 1.0 - b2 ** t,
                                         ops._stream()), "ofq_adamw_multi")
        return loss
