"""Data-parallel QAT over the GPUs of one node: one process per GPU, RCCL (torch.distributed backend "nccl")
over xGMI, gradients all-reduced in a few large flat buckets that are launched while backward is still running.

This replaces the reference's `NativeDDP(model, device_ids=[local_rank])` (train.py:727) and its constructor
broadcast (which is what makes rank 0's data-dependent LSQ step sizes, created by setup_alpha at train.py:657,
the global ones).  Design notes for MI355X:
  * xGMI is point-to-point (7 links x ~153 GB/s per GPU): a ring all-reduce is per-link bound, so few, large
    messages beat many small ones.  DeiT-S QKR has 90.8 MB of fp32 gradients: 4 buckets of ~24 MB by default.
  * each bucket is one flat buffer: when its last gradient lands the gradients are packed with ONE multi-tensor copy
    (not one accumulate kernel per parameter), reduced in place (RCCL AVG), and p.grad is re-pointed at the slices,
    so AdamW reads the averaged values with no unpack.  The large weight gradients never take part in that copy: the
    split-K reduce of their dW GEMM writes straight into the bucket slice (grad_slot below; 78 % of DeiT-S's gradient
    bytes: the pack kernel went from 0.32 to 0.12 ms per step).
  * buckets are filled in reverse parameter order (heads and last blocks finish first in backward) and each is
    launched from an autograd post-accumulate hook as soon as its last gradient lands.  The first synchronised backward
    records the order in which the gradients really arrive and the buckets are rebuilt in that order (as torch's DDP
    does): the q / k weights of all QKR blocks get their gradients from ONE node at the very end of backward
    (functional.AllWqkFn), and in parameter order they would hold every bucket back until then.
  * buffers: DDP re-broadcasts rank 0's buffers before every forward (train.py:727, broadcast_buffers=True).  The only
    buffers on this path are the stem quantiser's data-latched `signed` flags (lsq.py:310, 338-355), which go 0 -> 1 once
    and never back: sync_buffers() does the same broadcast before each forward until rank 0 holds them at 1, after
    which the broadcast is the identity and is skipped (no per-step collective, no per-step host sync).
  * a parameter that takes no part in a step: DDP (find_unused_parameters=False, the reference's setting) raises;
    so does finish_gradient_sync().
StatsQ statistics need no collective: s = 2*mean|W| is a pure function of replica-identical weights
(SURVEY.md §2.3); `check_statsq_consistency` verifies exactly that with one tiny all-reduce.
"""
import sys
import weakref

import torch
import torch.distributed as dist


def _capturing(t):
    return t.is_cuda and torch.cuda.is_current_stream_capturing()


# Where a parameter's gradient will live once its bucket is packed: the kernels that produce the large weight gradients
# (functional.CodesLinearFn: the split-K reduce of the dW GEMM) write there directly, so the bucket pack copies only what
# is left (88 MB -> 20 MB per DeiT-S step).  id(param) -> (weakref(param), bucket view); set by DataParallel._build_buckets.
_GRAD_SLOTS = {}


def grad_slot(param):
    """A fresh alias of `param`'s slice of its gradient bucket, or None (no synchronised DataParallel owns it, or the
    parameter already holds a gradient this step -- a second backward must accumulate, not overwrite)."""
    if param is None or param.grad is not None:
        return None
    hit = _GRAD_SLOTS.get(id(param))
    if hit is None or hit[0]() is not param:
        return None
    if hit[2][0]:
        # a second node asks for the same leaf's slot in one backward (a module called twice, weight tying, checkpointing):
        # both would write the slice and autograd would then add two aliases of one memory -- only the first writer gets
        # the slot, the others produce ordinary gradients that autograd accumulates
        return None
    hit[2][0] = True
    return hit[1].detach()        # a new tensor object on the same memory: autograd adopts it as .grad without a copy


class GradBucket:
    __slots__ = ("flat", "params", "pending", "work", "views", "flag")

    def __init__(self, flat, params):
        self.flat, self.params = flat, params
        self.pending = 0
        self.work = None


class DataParallel(torch.nn.Module):
    def __init__(self, module, process_group=None, bucket_mb=24.0, broadcast=True, force_sync=False, sync_statsq=False):
        super().__init__()
        self.module = module
        self.group = process_group
        self.world = dist.get_world_size(process_group) if dist.is_initialized() else 1
        # force_sync: run the hooks / collectives even in a one-rank group (exercises the RCCL path on a 1-GPU box)
        self.sync = self.world > 1 or (force_sync and dist.is_initialized())
        # sync_statsq: BASELINE.json's north_star names an all-reduce "of StatsQ statistics".  The reference has none and needs
        # none: s = 2 * mean|W| is a pure function of weights that are identical on every rank after the gradient all-reduce
        # (SURVEY.md 2.3 / 8e).  With this flag the per-row scales of every StatsQ quantiser are all-reduced (mean) right
        # behind the last gradient bucket of each step and ASSERTED to be a no-op (checked on the host every
        # `statsq_check_every` steps, so that the step itself gains no synchronisation point)
        self.sync_statsq = bool(sync_statsq) and self.sync
        self.statsq_check_every = 50
        self._statsq_pending = None
        self._steps = 0
        self._hooks = []
        self._bucket_mb = bucket_mb
        self._arrival = None               # parameter arrival order of the first synchronised backward (then: buckets rebuilt)
        self._rebuilt = False
        self._buffers_settled = False
        # pack_only: the gradient hooks pack each bucket (and point p.grad at its slices) but start no collective;
        # all_reduce_packed() then reduces every bucket eagerly.  This is how engine.GraphedTrainStep(mode="split") keeps
        # RCCL out of its captured graphs: [graph: forward + backward + packing] -> eager all-reduces -> [graph: optimiser].
        self.pack_only = False
        # on_packed(bucket index): called at the end of a pack_only bucket launch -- engine.GraphedTrainStep(mode="segmented")
        # ends the running stream capture there, so that the bucket's all-reduce can be issued between two sub-graphs
        self.on_packed = None
        # Work handles of the eager collectives issued since the last drain_collectives(): a stream capture is only started
        # once every one of them reports completion (engine.GraphedTrainStep._capture)
        self._eager_works = []
        self._pending_works = []
        if broadcast and self.sync:
            self.broadcast_parameters()
        self._build_buckets(bucket_mb)
        if self.sync_statsq:               # allocated here, not inside a stream capture (where it would be re-zeroed by every replay)
            p0 = next(self.module.parameters())
            self._statsq_pending = torch.zeros((), dtype=torch.float32, device=p0.device)

    # -- rank 0's parameters and buffers win (train.py:727: DDP's constructor broadcast)
    @torch.no_grad()
    def broadcast_parameters(self):
        tensors = [p.data for p in self.module.parameters()] + [b.data for b in self.module.buffers()]
        for dtype in {t.dtype for t in tensors}:
            group = [t for t in tensors if t.dtype == dtype]
            flat = torch.cat([t.reshape(-1) for t in group])
            self._broadcast(flat)
            off = 0
            for t in group:
                n = t.numel()
                t.copy_(flat[off:off + n].view_as(t))
                off += n
        self._forget_latches()

    def _broadcast(self, flat):
        w = dist.broadcast(flat, src=0, group=self.group, async_op=True)
        w.wait()
        self._keep_work(w)

    def _keep_work(self, w):
        """Remember the Work handle of an eager collective for drain_collectives() -- but only while it may still be in flight:
        a replayed step issues its collectives from the host on every step (modes "split" / "segmented") and nothing else
        would ever trim the list (zero_grad / finish_gradient_sync sit inside the captured graphs).  Handles that report
        completion are dropped as new ones arrive, so the list holds the last step's handles, not the run's, and no finished
        collective keeps its tensors (the StatsQ mean vector, the broadcast's flat copy of the weights) alive."""
        works = getattr(self, "_eager_works", None)
        if works is None:
            works = self._eager_works = []
        if len(works) >= 8:
            works[:] = [x for x in works if not x.is_completed()]
        works.append(w)

    def _latch_quantizers(self):
        return [m for m in self.module.modules() if hasattr(m, "sync_latch") and hasattr(m, "latched")]

    def _forget_latches(self):
        for q in self._latch_quantizers():
            q.sync_latch()

    @torch.no_grad()
    def sync_buffers(self):
        """DDP's per-forward buffer broadcast from rank 0, skipped once it has become the identity (see the module text)."""
        if not self.sync or self._buffers_settled:
            return
        bufs = [b.data for b in self.module.buffers()]
        if not bufs:
            self._buffers_settled = True
            return
        for dtype in {t.dtype for t in bufs}:
            group = [t for t in bufs if t.dtype == dtype]
            flat = torch.cat([t.reshape(-1) for t in group])
            self._broadcast(flat)
            off = 0
            for t in group:
                n = t.numel()
                t.copy_(flat[off:off + n].view_as(t))
                off += n
        self._forget_latches()
        latches = self._latch_quantizers()
        latch_bufs = {id(q.signed) for q in latches}
        # every rank now holds rank 0's values, so every rank takes the same decision
        self._buffers_settled = all(id(b) in latch_bufs for b in self.module.buffers()) and all(q.latched() for q in latches)

    def _build_buckets(self, bucket_mb, order=None):
        params = [p for p in self.module.parameters() if p.requires_grad]
        if order is not None:
            seen = {id(p) for p in order}
            params = [p for p in params if id(p) not in seen] + list(reversed(order))     # reversed again below
        for h in self._hooks:
            h.remove()
        self._hooks = []
        cap = int(bucket_mb * 1024 * 1024 / 4)
        groups, cur, cur_n = [], [], 0
        for p in reversed(params):                      # backward produces gradients roughly in reverse order
            cur.append(p)
            cur_n += p.numel()
            if cur_n >= cap:
                groups.append(cur)
                cur, cur_n = [], 0
        if cur:
            groups.append(cur)
        self.buckets = []
        self._bucket_of = {}
        for grp in groups:
            n = sum(p.numel() for p in grp)
            # one element more than the gradients: the bucket's step flag (this rank's stream-K error state when the bucket was
            # packed; the all-reduce averages it with the gradients, so afterwards every rank holds the same non-zero value when
            # ANY rank's step went wrong -- engine's step guard reads it, no extra collective)
            flat = torch.zeros(n + 1, dtype=grp[0].dtype, device=grp[0].device)
            b = GradBucket(flat, grp)
            b.flag = flat[n:n + 1]
            off = 0
            b.views = []
            for p in grp:
                b.views.append(flat[off:off + p.numel()].view_as(p))
                off += p.numel()
            self.buckets.append(b)
            for p, v in zip(grp, b.views):
                self._bucket_of[p] = b
                if self.sync:
                    _GRAD_SLOTS[id(p)] = (weakref.ref(p), v, [False])      # [taken in the current backward]
        if self.sync:
            for p in params:
                self._hooks.append(p.register_post_accumulate_grad_hook(self._on_grad))
        backend = dist.get_backend(self.group) if dist.is_initialized() else ""
        self._avg = backend == "nccl"                 # RCCL averages in the collective; gloo (CPU tests) sums, then divides
        self.zero_grad()

    def _reset(self):
        for b in self.buckets:
            b.pending = len(b.params)
            b.work = None
            if self.sync:
                for p in b.params:
                    hit = _GRAD_SLOTS.get(id(p))
                    if hit is not None:
                        hit[2][0] = False
        if self.sync and not self._rebuilt and self._arrival is None:
            self._arrival = []                        # record the first synchronised backward

    def _launch(self, b):
        """Pack the bucket's gradients into its flat buffer with one multi-tensor copy, start the asynchronous
        all-reduce (its stream waits for the kernels queued so far, then runs next to the rest of backward) and point
        every p.grad at its slice, which will hold the averaged value once the work completes."""
        fn = sys.modules.get(__package__ + ".functional")
        if fn is not None and fn.has_queued_work():
            fn.flush_dw()             # deferred weight-gradient GEMMs (functional.queue_dw) write into these gradients
        have = [(v, p.grad) for v, p in zip(b.views, b.params) if p.grad is not None]
        if len(have) < len(b.params):
            # torch's DDP with find_unused_parameters=False (the reference, train.py:727) fails here as well
            raise RuntimeError("ofq_amd DataParallel: %d parameter(s) of a gradient bucket took no part in this step "
                               "(no gradient); every trainable parameter must be used in the forward"
                               % (len(b.params) - len(have)))
        have = [(v, g) for v, g in have if g.data_ptr() != v.data_ptr()]       # written in place by its kernel (grad_slot)
        if have:
            torch._foreach_copy_([v for v, _ in have], [g for _, g in have])
        for v, p in zip(b.views, b.params):
            p.grad = v
        if b.flat.is_cuda:
            ops = sys.modules.get(__package__ + ".ops")
            if ops is not None:                       # flag <- 1.0 if a stream-K hand-off of this rank has timed out so far, else 0.0
                ops.step_guard(b.flat.device, flag_out=b.flag, set_guard=False)
                ops.LAUNCHES[0] -= 1                  # (not "work since the last cut" for engine's segmented capture)
        if self.pack_only:
            b.work = "packed"
            if self.on_packed is not None:
                self.on_packed(self.buckets.index(b))
            return
        b.work = self._all_reduce(b)

    def _all_reduce(self, b):
        """Start the asynchronous mean all-reduce of one packed bucket (RCCL: AVG inside the collective; gloo: divide, then SUM)."""
        if self._avg:
            w = dist.all_reduce(b.flat, op=dist.ReduceOp.AVG, group=self.group, async_op=True)
        else:
            b.flat.div_(self.world)
            w = dist.all_reduce(b.flat, op=dist.ReduceOp.SUM, group=self.group, async_op=True)
        if not _capturing(b.flat):
            self._keep_work(w)
        return w

    def all_reduce_bucket(self, i):
        """The all-reduce of bucket i of a pack_only step, on its own: engine.GraphedTrainStep(mode="segmented") issues it right
        after the sub-graph that packed the bucket, so that it runs on RCCL's stream next to the following sub-graph."""
        if self.sync:
            self._pending_works.append(self._all_reduce(self.buckets[i]))

    def wait_collectives(self):
        """The current stream waits for the collectives started by all_reduce_bucket (+ the StatsQ-scale check if it is on)."""
        if not self.sync:
            return
        if self.sync_statsq:
            self._statsq_all_reduce()
        for w in self._pending_works:
            w.wait()
        del self._pending_works[:]

    @torch.no_grad()
    def check_reduced_gradients(self):
        """After the bucket all-reduces of a step: every rank must hold the SAME bytes (an all-reduce hands every rank the same
        result).  One small MAX / MIN all-reduce pair over per-bucket checksums and a host sync: engine.GraphedTrainStep calls it
        for the first replays of a several-rank run, so that a mis-ordered collective (a reduce that ran before its bucket was
        packed, or raced with the optimiser) stops the run instead of training on different gradients per rank."""
        if not self.sync:
            return 0.0
        cs = torch.stack([torch.stack((b.flat.double().sum(), b.flat.double().abs().sum())) for b in self.buckets]).reshape(-1)
        hi, lo = cs.clone(), cs.clone()
        dist.all_reduce(hi, op=dist.ReduceOp.MAX, group=self.group)
        dist.all_reduce(lo, op=dist.ReduceOp.MIN, group=self.group)
        dev = float((hi - lo).abs().max())
        if dev != 0.0 or not bool(torch.isfinite(cs).all()):
            raise RuntimeError("ofq_amd DataParallel: the ranks hold different (or non-finite) gradients after the all-reduce "
                               "(checksum spread %g): a collective ran out of order with the kernels around it" % dev)
        return dev

    def drain_collectives(self, timeout_s=30.0):
        """Before a stream capture: every eager collective issued so far must have COMPLETED as seen through its own Work
        handle (c10d's watchdog thread polls the same end events; a capture that starts while one of them is still in flight
        has ended the process on this stack).  Bounded: raises after timeout_s instead of sleeping blindly."""
        import time
        works, self._eager_works = self._eager_works, []
        t0 = time.monotonic()
        for w in works:
            w.wait()
        if works and torch.cuda.is_available():
            torch.cuda.synchronize()
        for w in works:
            while not w.is_completed():
                if time.monotonic() - t0 > timeout_s:
                    raise RuntimeError("ofq_amd DataParallel: a collective issued before the capture did not complete within %.0f s"
                                       % timeout_s)
                time.sleep(0.001)
        return len(works)

    def _on_grad(self, p):
        if self._arrival is not None:
            self._arrival.append(p)
        b = self._bucket_of[p]
        b.pending -= 1
        if b.pending == 0:
            self._launch(b)

    def zero_grad(self):
        """Gradients are produced into fresh tensors by autograd (no accumulate-into-view kernels) and packed per bucket
        in _launch, so 'zeroing' is dropping the references."""
        for b in self.buckets:
            for p in b.params:
                p.grad = None
        self._reset()

    def finish_gradient_sync(self):
        """Call after loss.backward() and before optimizer.step()."""
        if self.sync:
            for b in self.buckets:
                if b.work is None:                      # bucket with parameters that got no gradient this step
                    self._launch(b)
            if self.pack_only:                          # the collectives are the caller's (all_reduce_packed)
                self._reset()
                return
            if self.sync_statsq:
                self._statsq_all_reduce()
            for b in self.buckets:
                b.work.wait()
            if not self._rebuilt and self._arrival and _capturing(self.buckets[0].flat):
                # a first synchronised backward inside a stream capture cannot rebuild (allocation, copies): drop the
                # recording and take the next eager backward's (hooks do not fire during replays, so keeping it would
                # append a second pass and hand every parameter to _build_buckets twice)
                self._arrival = None
            if not self._rebuilt and self._arrival and not _capturing(self.buckets[0].flat):
                # same autograd graph on every rank => same arrival order on every rank => same buckets
                seen, order = set(), []
                for p in self._arrival:                # (first arrival counts, should a parameter ever be recorded twice)
                    if id(p) not in seen:
                        seen.add(id(p))
                        order.append(p)
                self._arrival, self._rebuilt = None, True
                grads = {id(p): p.grad for b in self.buckets for p in b.params}
                self._build_buckets(self._bucket_mb, order=order)
                for b in self.buckets:                # this step's averaged gradients move to the new slices
                    for v, p in zip(b.views, b.params):
                        v.copy_(grads[id(p)])
                        p.grad = v
                for b in self.buckets:
                    b.pending, b.work = len(b.params), None
                return
        self._reset()

    def all_reduce_packed(self):
        """The bucket all-reduces of a pack_only step, issued together (RCCL's stream waits for the packing kernels queued
        so far; the current stream then waits for the four collectives), then the StatsQ-scale check if it is on."""
        if not self.sync:
            return
        works = [self._all_reduce(b) for b in self.buckets]
        if self.sync_statsq:
            self._statsq_all_reduce()
        for w in works:
            w.wait()

    def _statsq_all_reduce(self):
        """all-reduce(mean) of every StatsQ scale vector of this step's forward; must leave them unchanged."""
        vecs = [m._s_dev.reshape(-1) for m in self.module.modules() if getattr(m, "_s_dev", None) is not None]
        if not vecs:
            return
        local = torch.cat(vecs)
        mean = local.clone()
        if self._avg:
            w = dist.all_reduce(mean, op=dist.ReduceOp.AVG, group=self.group, async_op=True)
            w.wait()
        else:
            w = dist.all_reduce(mean, op=dist.ReduceOp.SUM, group=self.group, async_op=True)
            w.wait()
            mean.div_(self.world)
        if not _capturing(local):
            self._keep_work(w)
        dev = (mean - local).abs().max()                 # stays on the device ...
        if self._statsq_pending is None:                 # (a persistent scalar: a captured step max-accumulates into the
            self._statsq_pending = torch.zeros((), dtype=dev.dtype, device=dev.device)     # same memory in every replay)
        self._statsq_pending.copy_(torch.maximum(self._statsq_pending, dev))
        self._steps += 1
        if self._steps % self.statsq_check_every == 0 and not _capturing(local):
            self.check_statsq_pending()                   # ... and is looked at every few steps (replays: by the host loop)

    def check_statsq_pending(self):
        """Host check of the deviations accumulated by _statsq_all_reduce (a synchronisation point).  Averaging identical
        values over a power-of-two number of ranks is exact; otherwise the mean may differ from the value in its last bit."""
        if self._statsq_pending is None:
            return 0.0
        dev = float(self._statsq_pending)
        self._statsq_pending.zero_()
        if dev > 0.0 and (self.world & (self.world - 1)) == 0:
            raise RuntimeError("ofq_amd DataParallel: StatsQ scales differ between ranks by %g -- the replicas' weights "
                               "have diverged" % dev)
        return dev

    def forward(self, *a, **k):
        return self.module(*a, **k)

    def release(self):
        """Detach from the module: remove the gradient hooks and this wrapper's direct-write slots (grad_slot would
        otherwise keep aliasing p.grad into the buckets of a wrapper that no longer synchronises anything)."""
        for h in self._hooks:
            h.remove()
        self._hooks = []
        for b in self.buckets:
            for p, v in zip(b.params, b.views):
                hit = _GRAD_SLOTS.get(id(p))
                # (only this wrapper's entries: a newer wrapper on the same module has replaced them with its own views)
                if hit is not None and hit[0]() is p and hit[1].data_ptr() == v.data_ptr():
                    del _GRAD_SLOTS[id(p)]

    def __del__(self):
        try:
            self.release()
        except Exception:  # noqa: BLE001  (interpreter shutdown)
            pass

    def flag_word_ptrs(self):
        """Device addresses of the buckets' step flags (see _build_buckets), for ops.step_guard."""
        return [b.flag.data_ptr() for b in self.buckets if b.flat.is_cuda]

    def gradient_bytes(self):
        return sum(b.flat.numel() * 4 for b in self.buckets)


@torch.no_grad()
def check_statsq_consistency(model, group=None):
    """All-reduce(max - min) of every StatsQ scale vector: must be exactly 0 when replicas are in sync."""
    vecs = [m._s_dev.reshape(-1) for m in model.modules() if getattr(m, "_s_dev", None) is not None]
    if not vecs or not dist.is_initialized():
        return 0.0
    flat = torch.cat(vecs)
    hi, lo = flat.clone(), flat.clone()
    dist.all_reduce(hi, op=dist.ReduceOp.MAX, group=group)
    dist.all_reduce(lo, op=dist.ReduceOp.MIN, group=group)
    return float((hi - lo).abs().max())


def reduce_tensor(t, world):
    """timm.utils.reduce_tensor (train.py:952): mean over ranks of a scalar metric."""
    rt = t.clone()
    dist.all_reduce(rt, op=dist.ReduceOp.SUM)
    return rt / world
