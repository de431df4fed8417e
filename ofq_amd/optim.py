"""AdamW on the HIP multi-tensor kernel (csrc/adamw.hip) -- SURVEY.md 8(f) rank 1.  Same update rule, hyper-parameters,
param-group handling and state-dict layout (`step`, `exp_avg`, `exp_avg_sq`, one `step` per tensor) as torch.optim.AdamW
(which the reference gets from timm's create_optimizer_v2, train.py:662), one launch per parameter group instead of a
chain of foreach kernels, and the CGA weight freeze (cga.py:953-1013) folded into the same pass via `set_frozen`.

hipGraph: a captured launch keeps its arguments, so inside a capture (engine.GraphedTrainStep) step() launches the
`_dev` form of the kernel, which reads lr and the bias corrections from device memory, and leaves the step counters
alone; `advance_for_replay()` -- called before every replay -- advances the counters on the host and stores the new
scalars (formed by the same host arithmetic as the eager path: bit-identical updates)."""
import ctypes as C

import numpy as np
import torch

from . import ops


class _GraphClass:
    """Parameters of one group that have taken the same number of steps: one captured launch, one device scalar block."""
    __slots__ = ("group", "params", "hyper")

    def __init__(self, group, params, hyper):
        self.group, self.params, self.hyper = group, params, hyper


class FusedAdamW(torch.optim.Optimizer):
    MAX_GRAPH_CLASSES = 32

    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2):
        if lr < 0 or eps < 0 or not 0 <= betas[0] < 1 or not 0 <= betas[1] < 1 or weight_decay < 0:
            raise ValueError("FusedAdamW: invalid hyper-parameter")
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay))
        self._frozen = {}            # id(param) -> {0,1} fp32 mask of the parameter's shape (CGA), or absent
        self._graph_classes = None   # list of _GraphClass while / after a capture
        self._hyper_pool = None

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

    def _ready(self, group):
        """Parameters of the group that take part in this step (have a gradient), state created on first use."""
        plist = [p for p in group["params"] if p.grad is not None]
        for p in plist:
            if not p.is_cuda or p.dtype != torch.float32 or not p.is_contiguous() or not p.grad.is_contiguous():
                raise RuntimeError("FusedAdamW: contiguous fp32 parameters on a HIP device only (no CPU fallback)")
            st = self.state[p]
            if not st:
                if torch.cuda.is_current_stream_capturing():
                    raise RuntimeError("FusedAdamW: optimizer state must exist before a capture (moments allocated inside a "
                                       "capture would be re-zeroed by every replay); run an eager step first")
                st["step"] = torch.tensor(0.0)
                st["exp_avg"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.preserve_format)
        return plist

    def _classes(self, plist):
        """Split by the number of steps taken so far: torch keeps one `step` per tensor, and a parameter that skipped a
        step (no gradient) must get its own bias corrections.  Normally one class."""
        out, seen = {}, {}
        for p in plist:
            st = self.state[p]["step"]
            t = seen.get(id(st))
            if t is None:
                t = seen[id(st)] = int(st.item() if torch.is_tensor(st) else st)
            out.setdefault(t, []).append(p)
        return out

    def _set_step(self, ps, t):
        """state['step'] = t for every parameter of the class.  The tensors of a class share ONE cpu tensor object (150
        fresh tensors per step cost more host time than the launch); it is split when the class splits."""
        first = self.state[ps[0]]["step"]
        shared = torch.is_tensor(first) and all(self.state[p]["step"] is first for p in ps)
        if shared:
            users = sum(1 for g in self.param_groups for p in g["params"] if self.state.get(p, {}).get("step") is first)
            shared = users == len(ps)
        if shared:
            first.fill_(float(t))
        else:
            obj = torch.tensor(float(t))
            for p in ps:
                self.state[p]["step"] = obj

    @staticmethod
    def _hyper(group, t):
        b1, b2 = group["betas"]
        h = np.empty(8, dtype=np.float32)
        ops._chk(ops.lib().ofq_adamw_hyper_pack(h.ctypes.data, float(group["lr"]), float(b1), float(b2), float(group["eps"]),
                                                float(group["weight_decay"]), 1.0 - b1 ** t, 1.0 - b2 ** t),
                 "ofq_adamw_hyper_pack")
        return h

    # ---------------------------------------------------------------------------------------------- hipGraph
    def begin_capture(self, device):
        self._graph_classes = []
        self._hyper_pool = torch.zeros(self.MAX_GRAPH_CLASSES * 8, dtype=torch.float32, device=device)

    def advance_for_replay(self):
        """Before every replay of a graph that holds this optimizer's step: advance the step counters of the captured
        classes and store their scalars (lr, bias corrections ...) where the captured launches read them."""
        if not self._graph_classes:
            raise RuntimeError("FusedAdamW.advance_for_replay: no captured step")
        vals = np.empty(len(self._graph_classes) * 8, dtype=np.float32)
        for i, gc in enumerate(self._graph_classes):
            st = self.state[gc.params[0]]["step"]
            t = int(st.item()) + 1
            self._set_step(gc.params, t)
            vals[8 * i:8 * i + 8] = self._hyper(gc.group, t)
        lib, stream = ops.lib(), ops._stream()
        for off in range(0, vals.size, 32):
            chunk = np.ascontiguousarray(vals[off:off + 32])
            ops._chk(lib.ofq_store_f32(self._hyper_pool.data_ptr() + 4 * off, chunk.ctypes.data, int(chunk.size), stream),
                     "ofq_store_f32")

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        lib = ops.lib()
        assert lib.ofq_adamw_tensor_entry_bytes() == 48
        capturing = torch.cuda.is_current_stream_capturing()
        if capturing and self._graph_classes is None:
            raise RuntimeError("FusedAdamW.step inside a stream capture needs begin_capture() first (engine.GraphedTrainStep)")
        for group in self.param_groups:
            plist = self._ready(group)
            if not plist:
                continue
            for t0, ps in self._classes(plist).items():
                tens = self._table(ps)
                # the device's step-guard word (engine's step writes it between the collectives and this call; None before the
                # first such step): while it is non-zero the launch updates nothing -- see ops.step_guard
                guard = ops.step_guard_ptr(ps[0].device) or None
                if capturing:
                    i = len(self._graph_classes)
                    if i >= self.MAX_GRAPH_CLASSES:
                        raise RuntimeError("FusedAdamW: more than %d (group, step) classes in one captured step" % i)
                    hyper = self._hyper_pool[8 * i:8 * i + 8]
                    self._graph_classes.append(_GraphClass(group, ps, hyper))
                    ops._chk(lib.ofq_adamw_multi_dev_g(tens.ctypes.data, len(ps), hyper.data_ptr(), guard, ops._stream()),
                             "ofq_adamw_multi_dev_g")
                    continue
                t = t0 + 1
                self._set_step(ps, t)
                b1, b2 = group["betas"]
                ops._chk(lib.ofq_adamw_multi_g(tens.ctypes.data, len(ps), float(group["lr"]), float(b1), float(b2),
                                               float(group["eps"]), float(group["weight_decay"]), 1.0 - b1 ** t, 1.0 - b2 ** t,
                                               guard, ops._stream()), "ofq_adamw_multi_g")
        return loss
