"""Step driver shared by train.py / cga.py / bench.py: model construction + surgery (get_qat_model,
train.py:386-426), setup_alpha (train.py:997-1010), optimizer with timm's no-weight-decay rule
(create_optimizer_v2, train.py:662), one training step (train.py:893-933) and the CGA hooks (cga.py:953-1013)."""
from types import SimpleNamespace

import os

import torch
import torch.nn as nn

from .deit import create_model
from .quantization.modules.utils import replace_module_by_qmodule_deit, replace_module_by_qmodule_swin
from .quantization.utils import KDLossSoftandHard
from . import ops
from . import functional as F_ofq

ACT_LAYER_MAPPINGS = {'relu': nn.ReLU, 'gelu': nn.GELU, 'prelu': nn.PReLU, 'rprelu': 'rprelu', 'None': 'None'}


def default_qmodules_swin(depths):
    """The name list of configs/swin_t_imagenet.attn_q.yml."""
    names, fi = ["features.0.0"], 1
    for si, d in enumerate(depths):
        for li in range(d):
            names += ["features.%d.%d.attn" % (fi, li), "features.%d.%d.mlp" % (fi, li)]
        fi += 1
        if si < len(depths) - 1:
            names.append("features.%d.reduction" % fi)
            fi += 1
    return names + ["head"]


def default_qmodules(depth):
    """The name list of configs/ours_imagenet_recipe.attn_q.yml:47-74."""
    names = ["patch_embed.proj"]
    for i in range(depth):
        names += ["blocks.%d.attn" % i, "blocks.%d.mlp" % i]
    return names + ["head", "head_dist"]


def get_qat_model(model, args):
    """train.py:386-426 — build per-module qconfigs from the flat args and swap the modules."""
    qconfigs = {}
    for m in args.qmodules:
        wcfg = {"mode": args.wq_mode if args.wq_enable else "Identity",
                "bit": args.wq_bitw if args.wq_bitw < 32 and args.aq_enable else "identity",
                "all_positive": False, "symmetric": not getattr(args, "wq_asym", False),
                "per_channel": args.wq_per_channel, "normalize_first": False,
                "learnable": getattr(args, "wq_clip_learnable", False)}
        acfg = {"enable": args.aq_enable if args.aq_enable else "Identity",
                "mode": args.aq_mode if args.aq_bitw < 32 and args.aq_enable else "identity", "bit": args.aq_bitw,
                "per_channel": args.aq_per_channel, "normalize_first": False,
                "learnable": getattr(args, "aq_clip_learnable", True)}
        qconfigs[m] = {"weight": wcfg, "act": acfg, "q_attn_dropout": getattr(args, "apply_q_attn_dropout", False),
                       "act_layer": ACT_LAYER_MAPPINGS[getattr(args, "act_layer", "gelu")]}
    if args.model_type == 'swin':
        return replace_module_by_qmodule_swin(model, qconfigs, pretrained_initialized=args.pretrained_initialized,
                                              qk_reparam=args.qk_reparam, qk_reparam_type=args.qk_reparam_type,
                                              boundaryRange=getattr(args, "boundaryRange", 0.005))
    if args.model_type != 'deit':
        raise ValueError("unknown model_type %r" % args.model_type)
    return replace_module_by_qmodule_deit(model, qconfigs, pretrained_initialized=args.pretrained_initialized,
                                          qk_reparam=args.qk_reparam, qk_reparam_type=args.qk_reparam_type,
                                          boundaryRange=getattr(args, "boundaryRange", 0.005))


def build_student(model_name="deit_small_distilled_patch16_224", wbits=2, abits=2, qk_reparam=True, qk_reparam_type=0,
                  num_classes=1000, depth=None, boundary_range=0.005, seed=42):
    torch.manual_seed(seed)                                     # train.py:258, :501
    kw = {"num_classes": num_classes}
    if depth is not None:
        kw["depth"] = depth
    model = create_model(model_name, **kw)
    is_swin = hasattr(model, "features")                        # torchvision-style Swin skeleton (swin.py)
    qmodules = (default_qmodules_swin([len(st) for st in model.features[1::2]]) if is_swin
                else default_qmodules(len(model.blocks)))
    args = SimpleNamespace(qmodules=qmodules, wq_mode="statsq", wq_enable=True,
                           wq_bitw=wbits, aq_enable=True, aq_mode="lsq", aq_bitw=abits, wq_per_channel=True,
                           aq_per_channel=True, aq_clip_learnable=True, wq_clip_learnable=False, act_layer="gelu",
                           model_type="swin" if is_swin else "deit", pretrained_initialized=True, qk_reparam=qk_reparam,
                           qk_reparam_type=qk_reparam_type, boundaryRange=boundary_range)
    return get_qat_model(model, args)


@torch.no_grad()
def setup_alpha(model, images):
    """train.py:997-1010: one eval-mode forward creates every lazily-initialised LSQ step size."""
    was_training = model.training
    model.eval()
    model(images)
    model.train(was_training)


def param_groups_weight_decay(model, weight_decay, skip=()):
    """timm optim_factory.param_groups_weight_decay (used by create_optimizer_v2): no decay for 1-D
    parameters, biases and the model's no_weight_decay() names."""
    skip = set(skip) | set(getattr(model, "no_weight_decay", lambda: set())())
    decay, no_decay = [], []
    for name, p in model.named_parameters():
        if not p.requires_grad:
            continue
        if p.ndim <= 1 or name.endswith(".bias") or name in skip:
            no_decay.append(p)
        else:
            decay.append(p)
    return [{"params": no_decay, "weight_decay": 0.0}, {"params": decay, "weight_decay": weight_decay}]


def make_optimizer(model, lr=5.47e-4, weight_decay=0.05, fused=None, hip=None):
    """AdamW with timm's no-decay split.  On a HIP device the multi-tensor kernel of csrc/adamw.hip (ofq_amd.optim.FusedAdamW,
    same rule / state dict as torch.optim.AdamW, CGA freeze folded in); hip=False selects torch's own fused AdamW."""
    groups = param_groups_weight_decay(model, weight_decay)
    on_dev = next(model.parameters()).is_cuda
    if hip is None:
        hip = on_dev and os.environ.get("OFQ_TORCH_ADAMW") is None
    if hip:
        from .optim import FusedAdamW
        return FusedAdamW(groups, lr=lr, weight_decay=weight_decay)
    if fused is None:
        fused = on_dev
    return torch.optim.AdamW(groups, lr=lr, weight_decay=weight_decay, fused=fused)


# -------------------------------------------------------------------------------------------- CGA (cga.py)
def cga_modules(model, qk_reparam=True, model_type='deit'):
    """cga.py:966-979: blocks.* modules whose name ends with fc1 | fc2 | .v | proj  (QKR), or qkv (plain);
    Swin (cga.py:957-964): no `blocks` filter, plus `reduction`."""
    out = []
    for k, v in model.named_modules():
        if model_type == 'swin':
            if qk_reparam and (k[-3:] == 'fc1' or k[-3:] == 'fc2' or k[-2:] == '.v' or k[-4:] == 'proj' or k[-9:] == 'reduction'):
                out.append((k, v))
            continue
        if 'blocks' not in k:
            continue
        if k[-3:] == 'fc1' or k[-3:] == 'fc2' or k[-4:] == 'proj' or (k[-2:] == '.v' if qk_reparam else k[-3:] == 'qkv'):
            out.append((k, v))
    return out


class CGAHooks:
    """freeze-outside-boundary gradient mask + weight restore around optimizer.step() (cga.py:953-1013)."""

    def __init__(self, model, wbits, boundary_range=0.005, qk_reparam=True, model_type='deit'):
        self.mods = cga_modules(model, qk_reparam, model_type)
        self.bits, self.br = wbits, boundary_range
        self.state = {}

    @torch.no_grad()
    def before_step(self, optimizer=None):
        """optimizer: a FusedAdamW takes the masks itself (gradient mask + weight restore inside its one pass); any other
        optimizer gets the reference's three-kernel sequence around its step."""
        self._folded = optimizer is not None and hasattr(optimizer, "set_frozen")
        ws = [m.weight.data for _, m in self.mods]
        multi = bool(ws) and all(w.is_cuda and w.dim() == 2 and w.is_contiguous() for w in ws)
        if multi:      # every mask in three launches (persistent mask / scratch buffers)
            self._masks, self._ranges = ops.cga_freeze_mask_multi(ws, self.bits, self.br, getattr(self, "_masks", None),
                                                                  getattr(self, "_ranges", None))
        for i, (k, m) in enumerate(self.mods):
            frz = self._masks[i] if multi else ops.cga_freeze_mask(m.weight.data, self.bits, self.br)   # cga.py:960
            if self._folded:
                optimizer.set_frozen(m.weight, frz)
                self.state[k] = (frz, None)
            else:
                saved = ops.cga_mask_grad_save(m.weight.grad, m.weight.data, frz)  # :962-964
                self.state[k] = (frz, saved)

    @torch.no_grad()
    def after_step(self, optimizer=None):
        for k, m in self.mods:
            frz, saved = self.state[k]
            if saved is not None:
                ops.cga_restore(m.weight.data, frz, saved)                         # :994-997
        if getattr(self, "_folded", False) and optimizer is not None:
            optimizer.clear_frozen()
        self.state.clear()


WEIGHT_CODE_CACHE = os.environ.get("OFQ_NO_WEIGHT_CODE_CACHE", "0") != "1"


def _statsq_modules(model):
    lst = getattr(model, "_ofq_statsq_list", None)
    if lst is None:
        from .quantization.quantizer.statsq import StatsQuantizer
        lst = [m for m in model.modules() if isinstance(m, StatsQuantizer)]
        model._ofq_statsq_list = lst
    return lst


def refresh_weight_codes(model):
    """StatsQ operands (scale, int8 codes, transposed bf16 codes, offset row-dots) of every quantised linear layer whose
    weight is a leaf Parameter, in one or two launches (ofq_statsq_codes_multi) instead of one launch inside each forward.
    Valid until invalidate_weight_codes(): train_step() brackets exactly the forward/backward of one step with the two,
    so nothing can observe operands of stale weights (the optimizer, CGA restore and checkpoint loads all run outside)."""
    from . import ops
    todo = []
    for q in _statsq_modules(model):
        la = q._last_args
        if la is not None and la[0] is not None and la[0].is_cuda and la[0].dim() == 2 and la[0].is_contiguous():
            todo.append((q, la))
    if not todo:
        return 0
    res = ops.statsq_codes_multi([(la[0].detach(), la[1], None if la[2] is None else la[2].detach(), la[3]) for _, la in todo])
    for (q, la), r in zip(todo, res):
        q._pre = (la[0], la[2], la[3], r)
    return len(todo)


def invalidate_weight_codes(model):
    for q in _statsq_modules(model):
        q._pre = None


def _step_body(model, optimizer, images, target, soft_target, loss_fn, dp, cga):
    """Everything of one step that touches the device (train.py:893-933); GraphedTrainStep captures exactly this."""
    loss = _step_compute(model, optimizer, images, target, soft_target, loss_fn, dp)
    _step_update(optimizer, dp, cga, loss=loss)
    return loss


def _step_compute(model, optimizer, images, target, soft_target, loss_fn, dp):
    """zero_grad, StatsQ refresh, forward, loss, backward (with dp: the bucket hooks -- collectives, or packing only)."""
    if dp is not None:
        dp.zero_grad()
    else:
        optimizer.zero_grad(set_to_none=True)
    if WEIGHT_CODE_CACHE:
        refresh_weight_codes(model)
        F_ofq.STEP_CACHE_ACTIVE = True
    try:
        out, _ = (dp or model)(images)
    finally:
        if WEIGHT_CODE_CACHE:
            F_ofq.STEP_CACHE_ACTIVE = False
            invalidate_weight_codes(model)      # the operands are captured by the autograd graph; the cache itself ends here
    loss = loss_fn(out, target, soft_target)
    F_ofq.DW_DEFER = True                       # weight-gradient GEMMs are queued and launched a block at a time
    F_ofq.begin_backward()
    if loss.is_cuda:
        ops.amax_begin(loss.device)             # the gradient tensors' maximum words (two-plane backward GEMMs): one fill per step
    try:
        loss.backward()
    except BaseException:
        F_ofq.drop_dw()
        raise
    finally:
        F_ofq.DW_DEFER = False
        ops.amax_end()
    F_ofq.flush_dw()
    F_ofq.assert_step_queues_empty()            # nothing parked, queued or deferred may outlive the backward pass
    if loss.is_cuda:
        ops.nt_sk_poison(loss.detach())         # a timed-out stream-K hand-off (corrupt dX) turns this step's loss into NaN
    return loss


def _step_update(optimizer, dp, cga, loss=None, guard_dp=None):
    """finish the gradient all-reduce, the step guard, [CGA mask], AdamW, [CGA restore] (train.py:927-933, cga.py:953-1013).

    The step guard (ops.step_guard; the reference has no counterpart -- torch's GEMMs cannot time out): between the collectives
    and the optimiser one tiny launch looks at this rank's stream-K error words and at the flag elements of the gradient buckets
    (guard_dp: the wrapper whose collectives the caller has issued itself; every bucket carries the flags of ALL ranks once it is
    reduced).  If any is set the loss of this step becomes NaN on EVERY rank and the AdamW launches that follow update nothing on
    EVERY rank; ops.nt_sk_poll then raises on every rank at the same step boundary."""
    if dp is not None:
        dp.finish_gradient_sync()
    if loss is not None and loss.is_cuda:
        gdp = dp if dp is not None else guard_dp
        ops.step_guard(loss.device, loss=loss.detach(), extra_words=gdp.flag_word_ptrs() if gdp is not None else ())
    if cga is not None:
        cga.before_step(optimizer)
    optimizer.step()
    if cga is not None:
        cga.after_step(optimizer)


def train_step(model, optimizer, images, target, soft_target, loss_fn=None, dp=None, cga=None):
    """One QAT step: student forward, KD loss, backward (+ bucketed all-reduce), [CGA mask], AdamW, [CGA restore]."""
    loss_fn = loss_fn or KDLossSoftandHard()
    if images.is_cuda:
        # raises when an earlier step's stream-K hand-off timed out (one rank: no host sync; several ranks: waits for the copy the
        # previous call queued -- a step behind the device at most -- so that every rank raises at the same step)
        ops.nt_sk_poll(images.device, wait=dp is not None and dp.sync and dp.world > 1)
    if dp is not None:
        dp.sync_buffers()                       # DDP's per-forward buffer broadcast (train.py:727, broadcast_buffers=True)
    return _step_body(model, optimizer, images, target, soft_target, loss_fn, dp, cga)


def _latch_quantizers(model):
    from .quantization.quantizer.lsq import LsqQuantizer4img
    return [(m, q) for m in model.modules() for q in [getattr(m, "input_quant_fn", None)] if isinstance(q, LsqQuantizer4img)]


class GraphedTrainStep:
    """train_step() with the device side of the step captured once in a hipGraph and replayed: one hipGraphLaunch instead of
    ~850 ctypes launches and ~20 ms of Python per step (the C ABI is capture-safe by construction: no allocation, no sync,
    no global state; torch's allocator serves the capture from a private pool, so every activation, gradient and workspace
    address is the same in every replay).

    What is NOT in the graph, because it changes from step to step on the host:
      * lr and the AdamW bias corrections -- FusedAdamW keeps them in device memory and advance_for_replay() rewrites
        them before each replay (same host arithmetic as the eager step: the update is bit-identical);
      * the batch: copied into the graph's static input buffers (a device-to-device copy; with alias_inputs=True the
        tensors of the capturing call are the static buffers and passing them again costs nothing -- bench.py);
      * DDP's per-forward buffer broadcast and the stem quantiser's signedness latch (lsq.py:338-355), a host decision:
        checked before the replay while the quantiser is still unsigned; if it flips, the step is re-captured.

    The first `warmup` calls run eagerly (real training steps: lazily created state -- optimizer moments, workspaces,
    CGA masks -- must exist before the capture); the next call captures and replays."""

    # what a capture next to a live process group does (see _capture); tools/capture_stress.py flips these to reproduce the
    # watchdog crash of round 4
    capture_error_mode_dp = "thread_local"
    drain_before_capture = True

    def __init__(self, model, optimizer, loss_fn=None, dp=None, cga=None, warmup=2, alias_inputs=False, mode="full"):
        if not hasattr(optimizer, "advance_for_replay"):
            raise RuntimeError("GraphedTrainStep needs ofq_amd.optim.FusedAdamW (per-step scalars in device memory)")
        if mode not in ("full", "split", "segmented"):
            raise ValueError("mode must be 'full', 'split' or 'segmented'")
        self.model, self.optimizer, self.dp, self.cga = model, optimizer, dp, cga
        # mode "split" (the default of bench.py / train.py with several ranks): captured compute, eager collectives --
        #   graph A = zero_grad + StatsQ refresh + forward + loss + backward + the packing of the gradient buckets,
        #   then the bucket all-reduces issued eagerly on RCCL's stream (DataParallel.all_reduce_packed),
        #   graph B = [CGA masks] + AdamW [+ CGA restore].
        # No collective is ever captured (RCCL inside a hipGraph has only been exercised with one rank here), the host issues
        # two graph launches and one collective per bucket per step instead of ~850 kernel launches, and the all-reduce is
        # exposed (90.8 MB of gradients over xGMI: well under a millisecond of a 22 ms step) instead of overlapped.
        # mode "segmented" (the default of bench.py / train.py with several ranks since round 5): graph A is cut at the bucket
        # boundaries -- the capture is ended and a new one begun inside the gradient hook that packs a bucket -- and bucket i's
        # all-reduce is issued eagerly right after sub-graph i has been launched: RCCL's stream waits for exactly that
        # sub-graph and reduces next to sub-graph i+1 (train.py:727's "all-reduce overlapped with backward" at ~2 host calls
        # per bucket); only the last bucket's collective is exposed.  The backward pass of the capture runs on the calling
        # thread (torch.autograd.set_multithreading_enabled(False)): a capture is ended by the thread that began it.
        self.mode = mode if (dp is not None and dp.sync) else "full"
        self.segments = []                       # [(graph, [bucket indices packed inside it])]
        self.verify_replays = 3                  # replays after which the ranks compare their reduced gradients (world > 1)
        self.graph_b = None
        self.loss_fn = loss_fn or KDLossSoftandHard()
        self.warmup = int(warmup)
        self.alias_inputs = bool(alias_inputs)   # True: the tensors of the capturing call ARE the static inputs (bench.py feeds
        self.calls = 0                           # the same resident batch every step); False: private copies, filled per call
        self.graph = None
        self.static = None
        self.loss = None
        self.captures = 0
        # the eager warm-up calls and the capture share ONE side stream: autograd nodes that outlive a step (AccumulateGrad
        # nodes pinned by gradient hooks, for instance) remember the stream they were created on, and a capture must never
        # touch the default stream
        self.stream = torch.cuda.Stream()
        self._static_grads = []
        self._latches = _latch_quantizers(model)

    def _unsigned_latches(self):
        return [(m, q) for m, q in self._latches if not q.latched()]

    def _capture(self, images, target, soft_target):
        dev = images.device
        if self.alias_inputs:
            self.static = (images, target, soft_target)
        else:
            self.static = tuple(t.clone() for t in (images, target, soft_target))
        images, target, soft_target = self.static
        self.optimizer.begin_capture(dev)
        self.optimizer.zero_grad(set_to_none=True)
        g = torch.cuda.CUDAGraph()
        torch.cuda.synchronize()
        mode = "global"
        if self.dp is not None and self.dp.sync:
            # With a process group alive, c10d's watchdog thread polls the end events of the collectives it has not retired yet
            # (hipEventQuery every 100 ms).  Under capture mode "global" every thread's event queries count as capture errors
            # while ANY thread captures, and on this stack that ends the process ("operation not permitted on an event last
            # recorded in a capturing stream", raised in the watchdog: tools/capture_stress.py reproduces it).  So: only this
            # thread's calls count (thread_local), and the capture starts only after every collective issued so far reports
            # completion through its own Work handle (bounded, raises on timeout -- no blind sleep).
            mode = self.capture_error_mode_dp
            if self.drain_before_capture:
                self.dp.drain_collectives()
        if self.mode == "segmented":
            self._capture_segmented(images, target, soft_target, mode)
            g = self.segments[0][0]
        elif self.mode == "split":
            self.dp.pack_only = True
            try:
                with torch.cuda.graph(g, stream=self.stream, capture_error_mode=mode):
                    self.loss = _step_compute(self.model, self.optimizer, images, target, soft_target, self.loss_fn, self.dp)
                    self.dp.finish_gradient_sync()         # packs what the hooks have not packed; starts nothing
                gb = torch.cuda.CUDAGraph()
                with torch.cuda.graph(gb, stream=self.stream, pool=g.pool(), capture_error_mode=mode):
                    _step_update(self.optimizer, None, self.cga, loss=self.loss, guard_dp=self.dp)
            finally:
                self.dp.pack_only = False
            self.graph_b = gb
        else:
            with torch.cuda.graph(g, stream=self.stream, capture_error_mode=mode):
                self.loss = _step_body(self.model, self.optimizer, images, target, soft_target, self.loss_fn, self.dp, self.cga)
            self.graph_b = None
        self.graph = g
        self.captures += 1
        # the gradient tensors the replays write (static addresses in the graph's pool)
        self._static_grads = [(p, p.grad) for p in self.model.parameters() if p.grad is not None]

    def _verify_reduction(self):
        """First replays of a several-rank run: all ranks must hold identical reduced gradients before the optimiser graph runs
        (DataParallel.check_reduced_gradients; a host sync, hence only `verify_replays` times)."""
        if self.verify_replays > 0 and self.dp is not None and self.dp.world > 1:
            self.verify_replays -= 1
            self.dp.check_reduced_gradients()

    def _capture_segmented(self, images, target, soft_target, capture_mode):
        """Graph A as a chain of sub-graphs cut at the bucket boundaries (see __init__), graph B as in split mode."""
        import gc
        dp = self.dp
        segs, state = [], {"g": None, "mark": 0, "carry": []}
        pool = [torch.cuda.graph_pool_handle()]        # one private pool for the whole chain (replayed in capture order)

        def begin():
            g = torch.cuda.CUDAGraph()
            g.capture_begin(pool=pool[0], capture_error_mode=capture_mode)
            state["g"], state["mark"] = g, ops.LAUNCHES[0]

        def end(buckets):
            state["g"].capture_end()
            segs.append((state["g"], buckets))
            state["g"] = None

        def on_packed(i):
            # the hook that has just packed bucket i (DataParallel._launch): close the running sub-graph here unless it is the
            # last bucket (whatever follows it -- the rest of backward, the loss check -- stays in its graph) or nothing has been
            # launched since the last cut (two buckets completed by one node: they share a sub-graph)
            state["carry"].append(i)
            packed = sum(len(b) for _, b in segs) + len(state["carry"])
            if packed < len(dp.buckets) and ops.LAUNCHES[0] > state["mark"]:
                with torch.cuda.stream(self.stream):
                    end(state["carry"])
                    state["carry"] = []
                    begin()

        gc.collect()
        torch.cuda.empty_cache()
        dp.pack_only, dp.on_packed = True, on_packed
        prev_mt = torch.autograd.is_multithreading_enabled()
        torch.autograd.set_multithreading_enabled(False)      # the hooks (and so capture_end / capture_begin) run on this thread
        try:
            with torch.cuda.stream(self.stream):
                begin()
                try:
                    self.loss = _step_compute(self.model, self.optimizer, images, target, soft_target, self.loss_fn, dp)
                    dp.finish_gradient_sync()                  # packs what the hooks have not packed; starts nothing
                    end(state["carry"])
                except BaseException:
                    if state["g"] is not None:                 # leave no stream in capture mode behind
                        try:
                            state["g"].capture_end()
                        except Exception:  # noqa: BLE001
                            pass
                    raise
            gb = torch.cuda.CUDAGraph()
            with torch.cuda.graph(gb, stream=self.stream, pool=pool[0], capture_error_mode=capture_mode):
                _step_update(self.optimizer, None, self.cga, loss=self.loss, guard_dp=self.dp)
        finally:
            torch.autograd.set_multithreading_enabled(prev_mt)
            dp.pack_only, dp.on_packed = False, None
        assert sorted(i for _, b in segs for i in b) == list(range(len(dp.buckets))), "a gradient bucket was never packed"
        self.segments, self.graph_b = segs, gb

    def __call__(self, images, target, soft_target):
        self.calls += 1
        if self.graph is None and self.calls <= self.warmup:
            cur = torch.cuda.current_stream()
            self.stream.wait_stream(cur)
            with torch.cuda.stream(self.stream):
                loss = train_step(self.model, self.optimizer, images, target, soft_target, self.loss_fn, self.dp, self.cga)
            cur.wait_stream(self.stream)
            return loss
        ops.nt_sk_poll(images.device, wait=self.dp is not None and self.dp.sync and self.dp.world > 1)
        if self.dp is not None:
            self.dp.sync_buffers()
        # the signedness latch of a still-unsigned image quantiser: decide on the host, as the eager forward would
        flipped = False
        unsigned = self._unsigned_latches()
        for m, q in unsigned:
            before = q.latched()
            q._latch(q.latch_input(images, getattr(m, "move_b4").bias))
            flipped |= q.latched() != before
        if self.dp is not None and self.dp.sync and self.dp.world > 1 and unsigned:
            # every rank must take the same decision (re-capture or replay): one rank's batch may flip the latch while the
            # others' do not -- and sync_buffers() hands rank 0's flag to everybody on the next call anyway.  Only while a
            # latch is still open (the decisions are taken together, so the ranks agree on that too): latch_input() syncs with
            # the host on those steps anyway, and afterwards the step gains no collective and no host sync from this
            import torch.distributed as dist
            f = torch.tensor([1.0 if flipped else 0.0], device=images.device)
            dist.all_reduce(f, op=dist.ReduceOp.MAX, group=self.dp.group)
            if float(f) > 0 and not flipped:
                for m, q in self._latches:
                    q.force_latch()
                flipped = True
        if self.graph is not None and (flipped or any(a.shape != b.shape or a.dtype != b.dtype
                                                      for a, b in zip((images, target, soft_target), self.static))):
            self.graph = None                              # clamp bounds / shapes are baked into the captured launches
        if self.graph is None:
            self._capture(images, target, soft_target)
        else:
            for dst, src in zip(self.static, (images, target, soft_target)):
                if dst.data_ptr() != src.data_ptr():
                    dst.copy_(src)
        self.optimizer.advance_for_replay()
        if self.mode == "segmented":
            for g, buckets in self.segments:
                g.replay()
                for i in buckets:                          # runs on RCCL's stream behind this sub-graph, next to the following one
                    self.dp.all_reduce_bucket(i)
            self.dp.wait_collectives()
            self._verify_reduction()
            self.graph_b.replay()
        else:
            self.graph.replay()
            if self.graph_b is not None:
                self.dp.all_reduce_packed()
                self._verify_reduction()
                self.graph_b.replay()
        if self._static_grads and self._static_grads[0][0].grad is not self._static_grads[0][1]:
            for p, g in self._static_grads:            # an eager step in between re-pointed p.grad: show the replay's
                p.grad = g
        return self.loss
