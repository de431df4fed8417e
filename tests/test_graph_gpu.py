"""GPU tests of the captured training step (engine.GraphedTrainStep): the hipGraph replay of a step must be the same
function, bit for bit, as the eager step (train.py:893-933 semantics) -- same losses, same parameters, same optimizer
state -- including the things a capture could silently freeze: the AdamW bias corrections and learning rate (which
change every step), the CGA masks (recomputed from the weights every step, cga.py:953-1013), the bucketed RCCL
all-reduce of the data-parallel wrapper, and the stem quantiser's data-dependent signedness latch (lsq.py:338-355)."""
import copy
import os

import numpy as np
import pytest
import torch

from util import rel_err

pytestmark = pytest.mark.gpu


def _tiny(qk_reparam=True, qk_reparam_type=0, depth=2, seed=0, model="deit_tiny_distilled_patch16_224", bits=3):
    from ofq_amd import engine
    torch.manual_seed(seed)
    m = engine.build_student(model, bits, bits, qk_reparam=qk_reparam, qk_reparam_type=qk_reparam_type, depth=depth).cuda()
    return m


def _batch(B=4, seed=1, nonneg=False):
    g = torch.Generator(device="cuda").manual_seed(seed)
    imgs = torch.randn(B, 3, 224, 224, device="cuda", generator=g)
    if nonneg:
        imgs = imgs.abs() + 0.01
    tgt = torch.randint(0, 1000, (B,), device="cuda", generator=g)
    soft = torch.randn(B, 1000, device="cuda", generator=g)
    return imgs, tgt, soft


def _state(model, opt):
    ps = [p.detach().clone() for p in model.parameters()]
    st = []
    for g in opt.param_groups:
        for p in g["params"]:
            s = opt.state.get(p)
            if s:
                st.append((float(s["step"]), s["exp_avg"].clone(), s["exp_avg_sq"].clone()))
    return ps, st


def _same(a, b):
    (pa, sa), (pb, sb) = a, b
    assert len(pa) == len(pb) and len(sa) == len(sb)
    for x, y in zip(pa, pb):
        assert torch.equal(x, y)
    for (t1, m1, v1), (t2, m2, v2) in zip(sa, sb):
        assert t1 == t2 and torch.equal(m1, m2) and torch.equal(v1, v2)


def _lr_at(i):
    return 5e-4 * (1.0 - 0.07 * i)          # a schedule: the captured AdamW launches must pick up every change


def _run(base, steps, graphed, batches, cga=False, dp_factory=None, qk_reparam=True, mode="full"):
    from ofq_amd import engine
    from ofq_amd.quantization.utils import KDLossSoftandHard
    model = copy.deepcopy(base).train()
    dp = dp_factory(model) if dp_factory else None
    opt = engine.make_optimizer(model, lr=_lr_at(0), weight_decay=0.05)
    hooks = engine.CGAHooks(model, 3, 0.05, qk_reparam=qk_reparam) if cga else None
    loss_fn = KDLossSoftandHard()
    gs = engine.GraphedTrainStep(model, opt, loss_fn, dp=dp, cga=hooks, warmup=2, mode=mode) if graphed else None
    losses = []
    for i in range(steps):
        for g in opt.param_groups:
            g["lr"] = _lr_at(i)
        imgs, tgt, soft = batches[i % len(batches)]
        if graphed:
            loss = gs(imgs, tgt, soft)
        else:
            loss = engine.train_step(model, opt, imgs, tgt, soft, loss_fn, dp=dp, cga=hooks)
        losses.append(float(loss.detach()))
    return losses, _state(model, opt), gs, model


def test_graph_replay_equals_eager_steps_bit_for_bit():
    """6 steps (2 eager warm-up calls, 1 capture + replay, 3 replays) with a changing learning rate and two alternating
    batches against 6 eager steps: losses, parameters and AdamW state identical."""
    from ofq_amd import engine
    base = _tiny()
    b0, b1 = _batch(seed=1), _batch(seed=2)
    engine.setup_alpha(base, b0[0])
    le, se, _, _ = _run(base, 6, False, [b0, b1])
    lg, sg, gs, model = _run(base, 6, True, [b0, b1])
    assert gs.captures == 1 and gs.graph is not None
    assert le == lg, (le, lg)
    _same(se, sg)
    # the gradients a caller sees after a replay are the ones the replay produced (static tensors of the graph)
    assert all(p.grad is not None for p in model.parameters() if p.requires_grad)


def test_deferred_weight_gradients_equal_immediate():
    """engine's step queues the dW GEMMs of the linear layers and launches them grouped (functional.queue_dw): every
    gradient must be the one the immediate launches give (same products; the split-K factor differs, so 1e-6 instead
    of bit equality), none may be a copy taken before the deferred kernel ran, and three steps must track each other."""
    from ofq_amd import engine
    import ofq_amd.functional as Fn
    base = _tiny(depth=3)
    b0 = _batch(seed=5)
    engine.setup_alpha(base, b0[0])
    res = {}
    for grouped in (False, True):
        Fn.DW_GROUP = grouped
        try:
            model = copy.deepcopy(base).train()
            opt = engine.make_optimizer(model, lr=1e-4, weight_decay=0.05)
            # one step with the optimiser update switched off: gradients of identical weights
            for g in opt.param_groups:
                g["lr"] = 0.0
                g["weight_decay"] = 0.0
            engine.train_step(model, opt, *b0)
            grads = {n: p.grad.detach().clone() for n, p in model.named_parameters() if p.grad is not None}
            for g in opt.param_groups:
                g["lr"] = 1e-4
            losses = [float(engine.train_step(model, opt, *b0).detach()) for _ in range(3)]
            res[grouped] = (grads, losses)
        finally:
            Fn.DW_GROUP = True
    ga, gb = res[False][0], res[True][0]
    assert ga.keys() == gb.keys()
    queued = 0
    for n in ga:
        a, b = ga[n].double(), gb[n].double()
        assert torch.isfinite(b).all(), n
        err = float((a - b).abs().max() / (a.abs().max() + 1e-30))
        assert err < 2e-6, (n, err)
        queued += int(n.endswith("fc2.weight"))
    assert queued == 3
    for x, y in zip(res[False][1], res[True][1]):
        assert abs(x - y) < 1e-4 * abs(x), (res[False][1], res[True][1])


@pytest.mark.parametrize("variant", ["deit_qkr", "deit_plain", "swin_qkr", "deit_qkr_frozen"])
def test_deferred_second_stage_sums_equal_immediate(variant):
    """Inside engine's step the second-stage reductions of the quantiser / LayerNorm backward kernels (d step, d offset,
    d gamma, d beta) are queued in the library and launched forty at a time (ops.deferred_sums, ofq_sum_flush).  Every
    gradient of the model must equal, bit for bit, the one the immediate launches give -- a gradient that was read before
    its reduction ran (accumulated into, copied, viewed) would show up here as garbage -- for the QKR DeiT, the plain
    DeiT and the Swin student, over a first step and a second one taken from the updated weights."""
    from ofq_amd import engine, ops
    import ofq_amd.functional as Fn
    if variant == "swin_qkr":
        torch.manual_seed(0)
        base = engine.build_student("swin_t", 3, 3, qk_reparam=True).cuda()
    else:
        base = _tiny(qk_reparam=variant != "deit_plain", depth=2)
    b0 = _batch(seed=9)
    engine.setup_alpha(base, b0[0])
    if variant == "deit_qkr_frozen":
        # parameters that do not take a gradient: autograd drops what the backward returns for them at once, so nothing
        # may be queued to be written there later (the memory belongs to somebody else by then)
        for n, p in base.named_parameters():
            if "blocks.0.norm" in n or ("blocks.1" in n and n.endswith(".s")) or ("move" in n and "blocks.1.attn" in n):
                p.requires_grad_(False)
    res, queued = {}, {}
    real_flush = ops.sum_flush
    for defer in (False, True):
        Fn.SUM_DEFER = defer
        ops._WS_POISON = defer            # the private partial buffers of the queued calls start as NaNs: nothing unwritten is summed
        seen = [0]

        def counting_flush():
            seen[0] += ops.lib().ofq_sum_pending()
            real_flush()
        ops.sum_flush = counting_flush
        try:
            model = copy.deepcopy(base).train()
            opt = engine.make_optimizer(model, lr=1e-4, weight_decay=0.05)
            out = []
            for _ in range(2):
                engine.train_step(model, opt, *b0)
                out.append({n: p.grad.detach().clone() for n, p in model.named_parameters() if p.grad is not None})
            res[defer], queued[defer] = out, seen[0]
        finally:
            Fn.SUM_DEFER = True
            ops._WS_POISON = False
            ops.sum_flush = real_flush
    assert queued[False] == 0 and queued[True] >= 20, queued
    for step in range(2):
        ga, gb = res[False][step], res[True][step]
        assert ga.keys() == gb.keys()
        for n in ga:
            assert torch.equal(ga[n], gb[n]), (variant, step, n, float((ga[n] - gb[n]).abs().max()))


def test_a_block_used_twice_in_one_step_gets_the_sum_of_both_gradients():
    """Tied weights / a module called twice: every leaf of the second block receives TWO gradients in one backward pass.
    The deferred launches (queued dW GEMMs, queued second-stage sums) hand autograd tensors that are written later, which is
    only sound for a gradient that is adopted, never for one that is added to another: the second gradient of a leaf must
    take the immediate path (functional._claim).  Gradients inside engine's step == gradients of a plain backward pass."""
    from ofq_amd import engine
    from ofq_amd.quantization.utils import KDLossSoftandHard
    import ofq_amd.functional as Fn
    base = _tiny(qk_reparam=True, depth=2)
    b0 = _batch(seed=5)
    engine.setup_alpha(base, b0[0])
    base.blocks[1] = base.blocks[0]                      # the same parameters, quantisers and biases, twice per forward
    loss_fn = KDLossSoftandHard()
    grads = {}
    for mode in ("plain", "step"):
        model = copy.deepcopy(base).train()
        assert model.blocks[1] is model.blocks[0]
        if mode == "plain":
            out, _ = model(b0[0])
            loss_fn(out, b0[1], b0[2]).backward()
        else:
            opt = engine.make_optimizer(model, lr=0.0, weight_decay=0.0)
            queued = [0]
            real = Fn.queue_dw

            def counting(*a, **k):
                queued[0] += 1
                return real(*a, **k)
            Fn.queue_dw = counting
            try:
                engine.train_step(model, opt, *b0)
            finally:
                Fn.queue_dw = real
            assert queued[0] == 1, queued           # fc2 (the tiny model's one groupable layer): first visit deferred, second not
        grads[mode] = {n: p.grad.detach().clone() for n, p in model.named_parameters() if p.grad is not None}
    assert grads["plain"].keys() == grads["step"].keys()
    for n, g in grads["plain"].items():
        assert rel_err(grads["step"][n].cpu(), g.cpu()) < 1e-5, n


def test_graph_replay_with_cga_hooks_equals_eager():
    """Config C5: QAttention_qkreparam_4_cga model, freeze masks recomputed from the weights inside every replay, mask and
    restore folded into the AdamW launch (cga.py:953-1013).  boundaryRange 0.05 so that a good share of weights freezes."""
    from ofq_amd import engine
    base = _tiny(qk_reparam_type=1)
    b0 = _batch(seed=3)
    engine.setup_alpha(base, b0[0])
    le, se, _, _ = _run(base, 5, False, [b0], cga=True)
    lg, sg, gs, _ = _run(base, 5, True, [b0], cga=True)
    assert gs.captures == 1
    assert le == lg
    _same(se, sg)


def test_graph_recaptures_when_the_image_quantiser_latches_signed():
    """Non-negative images keep LsqQuantizer4img unsigned (clamp 0..255); the first batch with a negative value flips the
    latch for good (lsq.py:338-355).  The clamp bounds are arguments of captured launches, so the step is re-captured."""
    from ofq_amd import engine
    base = _tiny()
    pos, neg = _batch(seed=4, nonneg=True), _batch(seed=5)
    engine.setup_alpha(base, pos[0])
    q = base.patch_embed.proj.input_quant_fn
    assert not q.latched()
    seq = [pos, pos, pos, pos, neg, neg, pos]
    le, se, _, me = _run(base, len(seq), False, seq)
    lg, sg, gs, mg = _run(base, len(seq), True, seq)
    assert gs.captures == 2
    assert mg.patch_embed.proj.input_quant_fn.latched() and me.patch_embed.proj.input_quant_fn.latched()
    assert le == lg
    _same(se, sg)


def _init_pg():
    import torch.distributed as dist
    if not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    return dist


def test_data_parallel_over_rccl_one_rank_equals_plain_step_and_graph():
    """DataParallel with its autograd hooks, flat buckets and RCCL all-reduce (one rank: AVG over one replica is the
    identity) against the plain step: gradients, losses and parameters bit for bit; then the same wrapper inside a captured
    step (RCCL collectives recorded in the hipGraph)."""
    from ofq_amd import engine, parallel
    dist = _init_pg()
    try:
        base = _tiny()
        b0 = _batch(seed=6)
        engine.setup_alpha(base, b0[0])

        def mk(model):
            return parallel.DataParallel(model, bucket_mb=1.0, force_sync=True)
        le, se, _, m_plain = _run(base, 4, False, [b0])
        ld, sd, _, m_dp = _run(base, 4, False, [b0], dp_factory=mk)
        assert le == ld
        _same(se, sd)
        for a, b in zip(m_plain.parameters(), m_dp.parameters()):
            if a.grad is not None:
                assert torch.equal(a.grad, b.grad)
        lg, sg, gs, _ = _run(base, 4, True, [b0], dp_factory=mk)
        assert gs.captures == 1
        assert le == lg
        _same(se, sg)
        # the large weight gradients are written into their bucket slices by the dW kernels (parallel.grad_slot): the pack
        # copy of a step moves only what is left (heads, stem, embeddings, q / k of the tiny model), and never a quantised linear layer's weight
        import copy
        from ofq_amd.quantization.utils import KDLossSoftandHard
        model = copy.deepcopy(base).train()
        dp = mk(model)
        opt = engine.make_optimizer(model)
        names = {id(p): n for n, p in model.named_parameters()}
        packed = []
        real_launch = dp._launch

        def launch(b):
            packed.extend(names[id(p)] for v, p in zip(b.views, b.params) if p.grad is not None and p.grad.data_ptr() != v.data_ptr())
            return real_launch(b)
        dp._launch = launch
        for _ in range(2):
            engine.train_step(model, opt, b0[0], b0[1], b0[2], KDLossSoftandHard(), dp=dp)
        total = sum(p.numel() for p in model.parameters() if p.requires_grad)
        copied = sum(dict(model.named_parameters())[n].numel() for n in set(packed))
        assert copied < 0.6 * total, (copied, total)
        assert not [n for n in packed if n.endswith(("fc1.weight", "fc2.weight", "proj.weight", "attn.v.weight"))
                    and "blocks" in n], packed
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("mode", ["segmented", "split"])
def test_split_graph_step_with_eager_collectives_equals_the_eager_step(mode):
    """engine.GraphedTrainStep(mode="segmented"), the several-rank default of bench.py / train.py since round 5: graph A cut into
    sub-graphs at the gradient-bucket boundaries, bucket i's all-reduce issued eagerly behind sub-graph i (so that it runs next
    to sub-graph i+1: train.py:727's overlap), then graph B; and mode="split", round 4's form: graph A = zero_grad + StatsQ
    refresh + forward + loss + backward + bucket packing, then the bucket all-reduces issued eagerly on RCCL's stream, then
    graph B = CGA masks + AdamW + restore (train.py:474, :727, :927-933; cga.py:953-1013).  One rank over RCCL (a 1-GPU box)
    against the eager DataParallel step and the plain step: losses, parameters and AdamW state bit for bit over seven steps
    with the CGA hooks on, a changing learning rate, alternating batches and the stem quantiser's signedness latch flipping in
    the fifth step (re-capture of both graphs); no collective may sit inside either graph."""
    from ofq_amd import engine, parallel
    dist = _init_pg()
    try:
        base = _tiny(qk_reparam_type=1)
        pos, neg = _batch(seed=4, nonneg=True), _batch(seed=5)
        engine.setup_alpha(base, pos[0])
        seq = [pos, pos, pos, pos, neg, neg, pos]

        def mk(model):
            return parallel.DataParallel(model, bucket_mb=1.0, force_sync=True, sync_statsq=True)
        lp, sp, _, _ = _run(base, len(seq), False, seq, cga=True)
        le, se, _, _ = _run(base, len(seq), False, seq, cga=True, dp_factory=mk)
        calls, events = [], []
        real, real_replay = dist.all_reduce, torch.cuda.CUDAGraph.replay

        def spy(t, *a, **k):
            calls.append(torch.cuda.is_current_stream_capturing())
            events.append("ar")
            return real(t, *a, **k)

        def spy_replay(self_):
            events.append("replay")
            return real_replay(self_)
        dist.all_reduce = spy
        torch.cuda.CUDAGraph.replay = spy_replay
        try:
            lg, sg, gs, mg = _run(base, len(seq), True, seq, cga=True, dp_factory=mk, mode=mode)
        finally:
            dist.all_reduce = real
            torch.cuda.CUDAGraph.replay = real_replay
        assert gs.mode == mode and gs.graph_b is not None and gs.captures == 2
        assert calls and not any(calls), "a collective was issued inside a stream capture"
        if mode == "segmented":
            # several sub-graphs, every bucket in exactly one of them, and in the last replayed step the first bucket's
            # all-reduce is on RCCL's stream BEFORE the last backward sub-graph is launched (overlap by construction)
            nseg = len(gs.segments)
            assert nseg >= 2 and sorted(i for _, b in gs.segments for i in b) == list(range(len(gs.dp.buckets)))
            last = events[-(nseg + 1 + len(gs.dp.buckets) + 1):]           # nseg replays + graph B + bucket ARs + StatsQ AR
            assert last.count("replay") == nseg + 1, last
            first_ar = last.index("ar")
            last_a_replay = [i for i, e in enumerate(last) if e == "replay"][-2]
            assert first_ar < last_a_replay, last
            assert last[-1] == "replay" and last[-2] == "ar"
        assert lp == le == lg, (lp, le, lg)
        _same(se, sg)
        _same(sp, sg)
        assert mg.patch_embed.proj.input_quant_fn.latched()
    finally:
        dist.destroy_process_group()


def test_cga_hooks_in_train_step_follow_the_reference_sequence():
    """engine.CGAHooks around engine.train_step (config C5: qk_reparam_type=1 model, mask / restore folded into the HIP
    AdamW) for three steps against the reference's own sequence (cga.py:953-1013) replayed through the oracle on CPU:
    loss.backward -> freeze_idx from the weights -> grad * (1 - idx) -> optimizer.step -> frozen weights restored.
    The frozen weights must come out of every step bit-identical to what went in; the losses follow the oracle's; the other
    parameters stay within the AdamW step bound of the oracle's (Adam normalises the gradient, so an element whose
    gradient is rounding noise moves by +-lr on either side: the 2-D weights are compared norm-wise, the rest by bound)."""
    import sys
    import ofq_oracle as O
    from functools import partial
    from types import SimpleNamespace
    import torch.nn as nn
    from ofq_amd import engine
    from ofq_amd.deit import DistilledVisionTransformer
    from ofq_amd.quantization.utils import KDLossSoftandHard
    torch.manual_seed(0)
    depth, dim, heads, ncls, B, bits, br = 2, 64, 2, 10, 4, 2, 0.05
    model = DistilledVisionTransformer(img_size=224, patch_size=16, embed_dim=dim, depth=depth, num_heads=heads, mlp_ratio=4,
                                       qkv_bias=True, num_classes=ncls, norm_layer=partial(nn.LayerNorm, eps=1e-6),
                                       act_layer=nn.GELU)
    with torch.no_grad():
        for p in model.parameters():
            if p.dim() >= 2:
                p.mul_(4.0)
            else:
                p.add_(0.05 * torch.randn_like(p))
    args = SimpleNamespace(qmodules=engine.default_qmodules(depth), wq_mode="statsq", wq_enable=True, wq_bitw=bits,
                           aq_enable=True, aq_mode="lsq", aq_bitw=bits, wq_per_channel=True, aq_per_channel=True,
                           model_type="deit", pretrained_initialized=True, qk_reparam=True, qk_reparam_type=1,
                           boundaryRange=br)
    model = engine.get_qat_model(model, args).cuda()
    img = torch.randn(B, 3, 224, 224, device="cuda")
    tgt = torch.randint(0, ncls, (B,), device="cuda")
    soft = torch.randn(B, ncls, device="cuda")
    engine.setup_alpha(model, img)
    model.train()
    lr, wd = 1e-4, 0.05
    # ---- the reference sequence on CPU through the oracle
    sd = {k: v.detach().cpu().clone() for k, v in model.state_dict().items()}
    leaves = {k: (v.requires_grad_(True) if v.dtype.is_floating_point and "clip_val" not in k and "signed" not in k else v)
              for k, v in sd.items()}
    names = {id(p): n for n, p in model.named_parameters()}
    groups = engine.param_groups_weight_decay(model, wd)
    ref_groups = [{"params": [leaves[names[id(p)]] for p in g["params"]], "weight_decay": g["weight_decay"]} for g in groups]
    ref_opt = torch.optim.AdamW(ref_groups, lr=lr, weight_decay=wd)
    cfg = dict(depth=depth, num_heads=heads, patch=16, wbits=bits, abits=bits, qkr=True)
    cga_names = [k + ".weight" for k, _ in engine.cga_modules(model, qk_reparam=True)]
    assert len(cga_names) == 4 * depth
    # ---- the product path
    opt = engine.make_optimizer(model, lr=lr, weight_decay=wd)
    hooks = engine.CGAHooks(model, bits, br, qk_reparam=True)
    loss_fn = KDLossSoftandHard()
    frozen_share = []
    for step in range(3):
        before = {n: dict(model.named_parameters())[n].detach().clone() for n in cga_names}
        loss = engine.train_step(model, opt, img, tgt, soft, loss_fn, cga=hooks)
        ref_opt.zero_grad(set_to_none=True)
        c, d = O.deit_forward(img.cpu(), leaves, cfg, training=True)
        lo = O.kd_loss_soft_and_hard(c, d, tgt.cpu(), soft.cpu())
        lo.backward()
        saved, idx = {}, {}
        for n in cga_names:                                                    # cga.py:958-964
            W = leaves[n]
            idx[n] = O.cga_freeze_idx(W.detach(), bits, br)
            W.grad = O.cga_mask_grad(W.grad, idx[n])
            saved[n] = W.detach().clone()
        ref_opt.step()
        with torch.no_grad():
            for n in cga_names:                                                # cga.py:994-997
                leaves[n].copy_(O.cga_restore(leaves[n].detach(), saved[n], idx[n]))
        lo = float(lo.detach())
        assert abs(float(loss.detach()) - lo) < 1e-4 * abs(lo), (step, float(loss.detach()), lo)
        for n, p in model.named_parameters():
            ref = leaves[n].detach()
            got = p.detach().cpu()
            if n in cga_names:
                frz = idx[n].bool()
                frozen_share.append(float(frz.float().mean()))
                # frozen weights: untouched, bit for bit (on both sides)
                assert torch.equal(got[frz], before[n].cpu()[frz]), (step, n)
                assert torch.equal(ref[frz], saved[n][frz])
                moved = got[~frz] != before[n].cpu()[~frz]
                assert float(moved.float().mean()) > 0.9, (step, n)           # ... and the others took their AdamW step
            assert float((got - ref).abs().max()) <= 2.5 * lr * (step + 1), (step, n)
            if got.dim() >= 2:
                err = float((got.double() - ref.double()).norm() / (ref.double().norm() + 1e-30))
                assert err < 1e-3, (step, n, err)
    assert 0.05 < sum(frozen_share) / len(frozen_share) < 0.95      # the mask froze a real share of the weights


def test_train_cli_two_steps_save_resume_and_validate(tmp_path):
    """train.py's entry point (ofq_amd.train_cli.main): reference flag names, a KD teacher loaded from --teacher-checkpoint,
    two epochs of two steps, checkpoint -> --resume continues with the same losses as an uninterrupted run (model, LSQ
    steps, AdamW state and epoch all restored, train.py:691-706), and validate() (train.py:1012-1083) equals the oracle's
    eval-mode logits on the same batches."""
    import io
    import contextlib
    import ofq_oracle as O
    from ofq_amd import train_cli
    from ofq_amd.deit import create_model
    torch.manual_seed(0)
    teacher = create_model("deit_tiny_distilled_patch16_224", num_classes=1000)
    tpath = str(tmp_path / "teacher.pth")
    torch.save({"state_dict": {"module." + k: v for k, v in teacher.state_dict().items()}}, tpath)   # DDP-style prefix
    common = ["--model", "deit_tiny_distilled_patch16_224", "--batch-size", "4", "--steps-per-epoch", "2", "--val-steps", "1",
              "--lr", "5e-4", "--weight-decay", "0.05", "--aq-enable", "--aq-mode", "lsq", "--aq-per-channel",
              "--aq_clip_learnable", "--aq-bitw", "3", "--wq-enable", "--wq-per-channel", "--wq-bitw", "3", "--wq-mode",
              "statsq", "--model_type", "deit", "--quantized", "--pretrained_initialized", "--use-kd", "--teacher",
              "deit_tiny_distilled_patch16_224", "--teacher-checkpoint", tpath, "--kd_hard_and_soft", "1", "--qk_reparam",
              "--qk_reparam_type", "0", "--log-interval", "1", "--warmup-epochs", "0"]

    def run(extra):
        buf = io.StringIO()
        with contextlib.redirect_stdout(buf):
            res = train_cli.main(common + extra)
        losses = [float(l.split("Loss:")[1].split()[0]) for l in buf.getvalue().splitlines() if l.startswith("Train:")]
        return res, losses
    full, losses_full = run(["--epochs", "2", "--output", str(tmp_path / "full")])
    assert len(losses_full) == 4
    first, losses_a = run(["--epochs", "1", "--output", str(tmp_path / "half")])
    # (the cosine schedule depends on --epochs: give the resumed run the full run's horizon)
    _, losses_b = run(["--epochs", "2", "--resume", str(tmp_path / "half" / "last.pth.tar"), "--output", str(tmp_path / "res")])
    assert len(losses_a) == 2 and len(losses_b) == 2
    assert losses_b[0] == pytest.approx(losses_full[2], rel=1e-6) and losses_b[1] == pytest.approx(losses_full[3], rel=1e-6)
    # unsupported flags do not pass silently
    with pytest.raises(SystemExit):
        train_cli.main(common + ["--epochs", "1", "--mixup", "0.8", "--mixup-mode", "elem"])
    # the recipe's augmentation flags (configs/ours_imagenet_recipe.attn_q.yml:18-26) drive the on-device input pipeline:
    # uint8 batches, mixup / cutmix with soft targets, normalisation, random erasing -- and the step still trains
    _, losses_aug = run(["--epochs", "1", "--mixup", "0.8", "--cutmix", "1.0", "--reprob", "0.25", "--remode", "pixel",
                         "--smoothing", "0.1"])
    assert len(losses_aug) == 2 and all(np.isfinite(l) and 5.0 < l < 20.0 for l in losses_aug)
    with pytest.raises(SystemExit):
        train_cli.main([a for a in common if a not in ("--teacher-checkpoint", tpath)] + ["--epochs", "1"])
    # validate() against the oracle's eval forward on the same synthetic validation batch
    model = full["model"]
    dev = next(model.parameters()).device
    vl = train_cli.SyntheticLoader(1, 4, 1000, dev, 42 + 1000)
    got = train_cli.validate(model, vl, 1, 0)
    sd = {k: v.detach().cpu() for k, v in model.state_dict().items()}
    cfg = dict(depth=12, num_heads=3, patch=16, wbits=3, abits=3, qkr=True)
    x, y = vl.pool[0]
    with torch.no_grad():
        logits = O.deit_forward(x.cpu(), sd, cfg, training=False)
    want_loss = float(torch.nn.functional.cross_entropy(logits, y.cpu()))
    # validate()'s numbers are the metrics of the model's own eval-mode logits ...
    model.eval()
    with torch.no_grad():
        own, _ = model(x)
    assert got["loss"] == pytest.approx(float(torch.nn.functional.cross_entropy(own, y)), rel=1e-6)
    assert got["top1"] == pytest.approx(100.0 * float((own.argmax(1) == y).float().mean()))
    # ... and they stay near the oracle's.  Only "near": the GPU sums |W| per row in fp64 (correctly rounded StatsQ scale),
    # torch-CPU in a vectorised fp32 cascade; the scales can differ by one ulp and a weight whose W / s lands within that
    # ulp of a rounding tie takes the neighbouring level (a fresh DeiT-T has about two such weights
    # among 5.4 M; tests/test_depth12_gpu.py measures the growth block by block), and twelve low-bit blocks amplify one flipped level to percents of the logits,
    # on either side.  Tie-free parity of the eval forward is pinned at 1e-3 by tests/test_modules_gpu.py (g7 eval_logits).
    dev_logits = float((own.cpu().double() - logits.double()).norm() / logits.double().norm())
    assert dev_logits < 0.5 and got["loss"] == pytest.approx(want_loss, rel=2e-2), (dev_logits, got["loss"], want_loss)


def test_fused_adamw_keeps_one_step_count_per_tensor():
    """torch.optim.AdamW advances `step` per tensor: a parameter without a gradient in some step keeps its own count and
    bias corrections (train.py:662 -> timm create_optimizer_v2 -> torch AdamW).  FusedAdamW must do the same -- it groups
    the tensors of a param group by their count and launches each class with its own scalars."""
    from ofq_amd.optim import FusedAdamW
    torch.manual_seed(0)
    a0, b0 = torch.randn(300, device="cuda"), torch.randn(17, 5, device="cuda")
    pa, pb = [torch.nn.Parameter(a0.clone()) for _ in range(2)], [torch.nn.Parameter(b0.clone()) for _ in range(2)]
    opts = [FusedAdamW([pa[0], pb[0]], lr=1e-2, weight_decay=0.05),
            torch.optim.AdamW([pa[1], pb[1]], lr=1e-2, weight_decay=0.05)]
    for step in range(5):
        ga, gb = torch.randn_like(a0), torch.randn_like(b0)
        for k, opt in enumerate(opts):
            pa[k].grad = ga.clone()
            pb[k].grad = None if step in (1, 2) else gb.clone()       # b sits out two steps
            opt.step()
    assert float(opts[0].state[pa[0]]["step"]) == 5 and float(opts[0].state[pb[0]]["step"]) == 3
    assert float(opts[1].state[pb[1]]["step"]) == 3
    for x, y in ((pa[0], pa[1]), (pb[0], pb[1])):
        assert float((x - y).abs().max()) < 2e-6 * float(y.abs().max())
    for key in ("exp_avg", "exp_avg_sq"):
        assert torch.allclose(opts[0].state[pb[0]][key], opts[1].state[pb[1]][key], rtol=1e-5, atol=1e-8)


def test_hip_teacher_forward_equals_the_fp32_deit():
    """The KD teacher's forward through the HIP kernels (ofq_amd.teacher.HipTeacher: fp32 MFMA GEMMs, fused add + LayerNorm,
    softmax kernel, erf-GELU) against (a) the oracle's restatement of the reference's fp32 distilled DeiT
    (deit_vision_transformer.py:85-164, deit.py:27-67) and (b) the stock PyTorch-ROCm forward of the same module, in the
    mode the reference runs its teacher in (training mode: cls logits distilled) and in eval mode.  fp32 against fp32 with
    different summation orders: 1e-5 of the logits' scale."""
    import ofq_oracle as O
    from ofq_amd.deit import create_model
    from ofq_amd.teacher import HipTeacher
    torch.manual_seed(0)
    model = create_model("deit_tiny_distilled_patch16_224", num_classes=1000).cuda()
    with torch.no_grad():
        for p in model.parameters():                       # random-init logits are ~0: give the comparison some scale
            if p.dim() >= 2:
                p.mul_(3.0)
        model.blocks[3].mlp.fc1.bias.normal_(0, 0.3)
    x = torch.randn(3, 3, 224, 224, device="cuda")
    sd = {k: v.detach().cpu() for k, v in model.state_dict().items()}
    for gemm in ("f32", "bf16x9", "bf16x6", "f16x4"):     # exact-fp32 MFMA; weights pre-split into bf16 planes (9 / 6 products); two fp16 planes each side
        teacher = HipTeacher(model, gemm=gemm)
        for training in (True, False):
            model.train(training)
            with torch.no_grad():
                stock, _ = model(x)
                got, _ = teacher(x)
                want = O.deit_fp32_forward(x.cpu(), sd, 12, 3, training=training)
            pairs = list(zip(got, stock, want)) if training else [(got, stock, want)]
            for g, s, w in pairs:
                scale = float(w.abs().max())
                assert float((g.cpu() - w).abs().max()) < 1e-5 * scale, gemm
                assert float((g - s).abs().max()) < 1e-5 * scale, gemm
    with pytest.raises(RuntimeError):
        teacher(x.cpu())


def test_bf16_plane_gemm_is_fp32_grade():
    """ofq_gemm_bf16x3x3_nt: both operands fp32, A split into bf16 planes in the kernel, B pre-split by ofq_split_f32_bf16x3.
    The planes reproduce the weights exactly; with all nine plane products the result is as close to the fp64 product as a
    true fp32 GEMM (torch / hipBLASLt) is, with the six leading ones within 2x of it."""
    from ofq_amd import ops
    g = torch.Generator(device="cuda").manual_seed(9)
    M, N, K = 1000, 392, 1536
    x = torch.randn(M, K, device="cuda", generator=g) * torch.pow(10.0, torch.randint(-3, 3, (M, 1), device="cuda", generator=g).float())
    W = torch.randn(N, K, device="cuda", generator=g) * 0.05
    b = torch.randn(N, device="cuda", generator=g)
    planes = ops.split_f32_bf16x3(W)
    assert torch.equal(planes[0].float() + planes[1].float() + planes[2].float(), W)          # exact split
    ref = x.double() @ W.double().t() + b.double()
    den = (x.double().abs() @ W.double().abs().t()) + 1e-30                                    # the GEMM error scale
    e_f32 = float(((torch.nn.functional.linear(x, W, b).double() - ref).abs() / den).max())
    e9 = float(((ops.gemm_bf16x3x3_nt(x, planes, b, products=9).double() - ref).abs() / den).max())
    e6 = float(((ops.gemm_bf16x3x3_nt(x, planes, b, products=6).double() - ref).abs() / den).max())
    assert e9 <= 1.5 * e_f32 + 1e-8 and e6 <= 3.0 * e_f32 + 1e-7, (e_f32, e9, e6)


def test_bf16_plane_gemm_wide_tile_equals_narrow_tile():
    """ofq_gemm_bf16x3x3_nt runs 128 x 384 double-buffered tiles for N >= 256 (gemm_bf16x3x3_wide_kernel) and 128 x 128 tiles
    below: same plane products in the same k order, so the wide result equals the narrow kernel's on 128-column slices of the
    weight, bit for bit -- nine and six products, ragged M / N, K = 384 and 1536."""
    from ofq_amd import ops
    g = torch.Generator(device="cuda").manual_seed(11)
    for (M, N, K) in [(1000, 392, 1536), (25344 // 8, 1152, 1024), (300, 384, 384)]:      # (shapes the launcher sends to the wide tile)
        x = torch.randn(M, K, device="cuda", generator=g)
        W = torch.randn(N, K, device="cuda", generator=g) * 0.05
        b = torch.randn(N, device="cuda", generator=g)
        planes = ops.split_f32_bf16x3(W)
        for products in (9, 6):
            wide = ops.gemm_bf16x3x3_nt(x, planes, b, products=products)
            for c0 in range(0, N, 128):
                c1 = min(N, c0 + 128)
                sl = planes[:, c0:c1].contiguous()
                narrow = ops.gemm_bf16x3x3_nt(x, sl, b[c0:c1].contiguous(), products=products)
                assert torch.equal(wide[:, c0:c1], narrow), (M, N, K, products, c0)


def test_bench_line_contract():
    """`python bench.py` prints exactly one JSON line with the driver's fields, the roofline block of the dominant kernel
    class (a bf16-split backward GEMM; at 128 images the dW GEMM) and the CPU baseline block (small step counts here: the numbers are not checked)."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--steps", "3", "--warmup", "1", "--batch-per-gpu", "16",
                        "--cpu-batch", "1", "--cpu-steps", "1", "--no-c1-baseline"], cwd=root, capture_output=True, timeout=900)
    assert r.returncode == 0, r.stderr.decode()[-2000:]
    lines = [l for l in r.stdout.decode().splitlines() if l.strip()]
    assert len(lines) == 1, lines
    d = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 3 and d["warmup"] == 1 and d["scaling"] == "weak" and d["vs_baseline"] is None
    assert d["unit"] == "images/s" and d["higher_is_better"] is True and "workload" in d["config"]
    # the precision trade of the backward GEMMs is part of the record: the mode in `config`, spelled out in `dtype`
    assert d["config"]["grad_planes"] in (2, 3) and d["dtype"].startswith("f32 (backward GEMM gradient operands as %d " % d["config"]["grad_planes"])
    assert abs(d["value"] - 16 * 3 / (d["ms_per_step"] * 3e-3)) < 1e-2 * d["value"]
    roof = d["roofline"]
    # (which class dominates depends on the batch: at 128 images the dW GEMM, at 16 any of the GEMM classes)
    assert roof["kernel"].split(" ")[0] in ("qgemm_bf16s_tn", "qgemm_bf16s_nt", "qgemm_bf16s_nn", "qgemm_i8_nt", "gemm_f32",
                                             "qgemm_i8_lsqbwd", "qattn_scores_softmax", "qgemm_bf16s_nt_wide",
                                             "qgemm_bf16s_tn_wide", "qgemm_bf16s_tn_wide_group", "qgemm_bf16s_tn_wide_stream",
                                             "qgemm_bf16s_nn_wide", "qattn_dp_softmax_bwd")
    assert "mfma_pipe_frac" in roof and roof["mfma_pipe_frac"] >= roof["frac"] - 1e-6
    assert roof["bound"] == "mfma" and roof["unit"] == "TFLOP/s"
    assert abs(roof["frac"] - roof["achieved"] / roof["peak"]) < 1e-3 and 0 < roof["frac"] < 1
    cb = d["cpu_baseline"]
    assert cb["kind"] == "port" and cb["cores"] >= 1 and cb["value"] > 0 and "sample" in cb



def _two_rank_worker(rank, world, port, errq):
    """One of two processes that SHARE the box's one GPU and talk over gloo (which moves device tensors through the host):
    the closest thing to a several-rank run a one-GPU box allows -- every rank-coordination path of the data-parallel step
    executes with a real peer (constructor broadcast, bucket all-reduces with world 2, the arrival-order rebuild, the latch
    agreement, captured compute with eager collectives)."""
    import traceback
    try:
        import torch.distributed as dist
        os.environ["MASTER_ADDR"] = "127.0.0.1"
        os.environ["MASTER_PORT"] = str(port)
        torch.cuda.set_device(0)
        dist.init_process_group("gloo", rank=rank, world_size=world)
        from ofq_amd import engine, parallel
        from ofq_amd.quantization.utils import KDLossSoftandHard
        base = _tiny(seed=rank)                              # different initial weights on every rank: rank 0's must win
        batches = [_batch(seed=20 + 2 * i + rank) for i in range(2)]      # rank-specific data
        engine.setup_alpha(base, batches[0][0])              # rank-specific LSQ steps, created lazily: rank 0's must win too
        loss_fn = KDLossSoftandHard()

        # (1) one synchronised step: the gradients are the mean over the ranks of the LOCAL gradients
        m_loc = copy.deepcopy(base).train()
        dp0 = parallel.DataParallel(copy.deepcopy(base).train(), bucket_mb=1.0)
        m_loc.load_state_dict(dp0.module.state_dict())       # rank 0's parameters, as the wrapper broadcast them
        out, _ = m_loc(batches[0][0])
        loss_fn(out, batches[0][1], batches[0][2]).backward()
        opt0 = engine.make_optimizer(dp0.module, lr=0.0, weight_decay=0.0)
        engine.train_step(dp0.module, opt0, *batches[0], loss_fn, dp=dp0)
        for (n, p), q in zip(m_loc.named_parameters(), dp0.module.parameters()):
            if p.grad is None:
                continue
            mean = p.grad.detach().clone()
            dist.all_reduce(mean)
            mean /= world
            assert rel_err(q.grad.cpu(), mean.cpu()) < 2e-6, (n, rel_err(q.grad.cpu(), mean.cpu()))
        dp0.release()

        # (2) five steps: eager DataParallel against captured compute + eager collectives, and rank against rank
        res = {}
        for mode in ("eager", "split", "segmented"):
            # (round 5 held the graph modes to EITHER of two eager runs and retried once: about one five-step run in twenty
            # ended with other last bits.  Round 6 traced that to ONE routine -- the row-dot of the attention prep launch returned
            # a wrong tq for 1-4 of 2376 rows about once in 400 calls whenever another process shared the GPU, two ranks or two
            # unrelated processes alike (tools/two_rank_trace.py) -- and fixed it: strict equality, one eager run.)
            model = copy.deepcopy(base).train()
            dp = parallel.DataParallel(model, bucket_mb=1.0)
            opt = engine.make_optimizer(model, lr=_lr_at(0), weight_decay=0.05)
            gs = engine.GraphedTrainStep(model, opt, loss_fn, dp=dp, warmup=2, mode=mode) if mode != "eager" else None
            losses = []
            for i in range(5):
                for g in opt.param_groups:
                    g["lr"] = _lr_at(i)
                b = batches[i % 2]
                loss = gs(*b) if gs is not None else engine.train_step(model, opt, *b, loss_fn, dp=dp)
                losses.append(float(loss.detach()))
            torch.cuda.synchronize()
            if gs is not None:
                assert gs.mode == mode and gs.graph_b is not None and gs.captures == 1
                assert gs.verify_replays == 0              # the ranks compared their reduced gradients after the first replays
                if mode == "segmented":
                    assert len(gs.segments) >= 2
            res[mode] = (losses, [p.detach().clone() for p in model.parameters()])
            flat = torch.cat([p.detach().reshape(-1) for p in model.parameters()])
            lo, hi = flat.clone(), flat.clone()
            dist.all_reduce(lo, op=dist.ReduceOp.MIN)
            dist.all_reduce(hi, op=dist.ReduceOp.MAX)
            assert torch.equal(lo, hi), "replicas diverged in mode " + mode
            dp.release()
        for mode in ("split", "segmented"):
            assert res["eager"][0] == res[mode][0], (mode, res["eager"][0], res[mode][0])
            assert all(torch.equal(a, b_) for a, b_ in zip(res["eager"][1], res[mode][1])), mode
        dist.barrier()
        dist.destroy_process_group()
    except BaseException:  # noqa: BLE001
        errq.put("rank %d:\n%s" % (rank, traceback.format_exc()))
        raise


def _two_rank_fault_worker(rank, world, port, errq):
    """A stream-K hand-off times out on rank 1 ONLY (fault injection): both ranks must see it in the same step -- NaN loss, no
    parameter, moment or step changed by that step's optimiser on either rank -- and both must raise at the same later call."""
    import traceback
    try:
        import numpy as np
        import torch.distributed as dist
        os.environ["MASTER_ADDR"] = "127.0.0.1"
        os.environ["MASTER_PORT"] = str(port)
        torch.cuda.set_device(0)
        dist.init_process_group("gloo", rank=rank, world_size=world)
        from ofq_amd import engine, ops, parallel
        from ofq_amd.quantization.utils import KDLossSoftandHard
        base = _tiny(seed=0)
        batches = [_batch(seed=40 + 2 * i + rank) for i in range(2)]
        engine.setup_alpha(base, batches[0][0])
        loss_fn = KDLossSoftandHard()
        dev = torch.device("cuda", 0)
        rs = np.random.RandomState(3)
        dy = torch.from_numpy(rs.randn(640, 256).astype(np.float32)).cuda()
        ks = torch.from_numpy((0.01 + rs.rand(256) * 0.09).astype(np.float32)).cuda()
        wT = ops.codes_transpose_bf16(torch.from_numpy((2 * rs.randint(-8, 8, (256, 384)) + 1).astype(np.int8)).cuda())
        scratch = torch.empty((640, 384), device="cuda")
        for mode in ("eager", "segmented"):
            model = copy.deepcopy(base).train()
            dp = parallel.DataParallel(model, bucket_mb=1.0)
            opt = engine.make_optimizer(model, lr=1e-3, weight_decay=0.05)
            gs = engine.GraphedTrainStep(model, opt, loss_fn, dp=dp, warmup=2, mode=mode) if mode != "eager" else None
            stream = gs.stream if gs is not None else torch.cuda.current_stream()
            with torch.cuda.stream(stream):                # the workspace (and its error word) of the stream the steps run on
                ops.qgemm_bf16s_nt_sk([(dy, wT, ks, 0.25)], scratch, wgs=3)
            torch.cuda.synchronize()
            step = (lambda b: gs(*b)) if gs is not None else (lambda b: engine.train_step(model, opt, *b, loss_fn, dp=dp))
            for i in range(4):
                assert torch.isfinite(step(batches[i % 2]))
            torch.cuda.synchronize()
            before = [p.detach().clone() for p in model.parameters()]
            moments = [opt.state[p]["exp_avg"].clone() for p in model.parameters() if p in opt.state]
            if rank == 1:
                with torch.cuda.stream(stream):
                    ops.nt_sk_inject_fault(dev, 1)
                    ops.qgemm_bf16s_nt_sk([(dy, wT, ks, 0.25)], scratch, wgs=3)
                    ops.nt_sk_inject_fault(dev, -1)
                torch.cuda.synchronize()
                assert ops.nt_sk_error(dev) != 0
            else:
                assert ops.nt_sk_error(dev) == 0
            raised_at = -1
            for i in range(4):
                try:
                    loss = step(batches[i % 2])
                    torch.cuda.synchronize()
                    assert torch.isnan(loss), (mode, rank, i, float(loss))       # on BOTH ranks, from the very step of the fault on
                except RuntimeError as e:
                    assert "stream-K hand-off timed out" in str(e), str(e)
                    raised_at = i
                    break
            assert raised_at == 2, (mode, rank, raised_at)                     # fault in step 0, seen by the poll of call 2 on both ranks
            torch.cuda.synchronize()
            assert all(torch.equal(a, p.detach()) for a, p in zip(before, model.parameters())), (mode, rank)
            assert all(torch.equal(a, opt.state[p]["exp_avg"]) for a, p in zip(moments, [q for q in model.parameters() if q in opt.state]))
            assert ops.nt_sk_error(dev) == 0                                   # re-zeroed by the poll that raised
            assert torch.isfinite(step(batches[0]))                            # and training can go on (from a checkpoint, in real life)
            torch.cuda.synchronize()
            flat = torch.cat([p.detach().reshape(-1) for p in model.parameters()])
            lo, hi = flat.clone(), flat.clone()
            dist.all_reduce(lo, op=dist.ReduceOp.MIN)
            dist.all_reduce(hi, op=dist.ReduceOp.MAX)
            assert torch.equal(lo, hi), "replicas diverged after the fault in mode " + mode
            dp.release()
        dist.barrier()
        dist.destroy_process_group()
    except BaseException:  # noqa: BLE001
        errq.put("rank %d:\n%s" % (rank, traceback.format_exc()))
        raise


def _spawn_two(worker, timeout=600):
    import socket
    import torch.multiprocessing as mp
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    errq = ctx.SimpleQueue()
    procs = [ctx.Process(target=worker, args=(r, 2, port, errq)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(timeout)
    msgs = []
    while not errq.empty():
        msgs.append(errq.get())
    for p in procs:
        if p.is_alive():
            p.kill()
            msgs.append("a rank was still running after %d s" % timeout)
    assert not msgs, "\n".join(msgs)
    assert all(p.exitcode == 0 for p in procs), [p.exitcode for p in procs]


def test_a_stream_k_timeout_on_one_rank_stops_the_optimiser_on_every_rank():
    """the error word used to reach the faulty rank's loss only, graph B applied the corrupt (reduced) gradients on every
    rank, and the faulty rank alone raised a step or two later while its peers blocked in their next collective.  Now every gradient
    bucket carries one flag element (this rank's error state when the bucket was packed; averaged by the bucket's own all-reduce),
    ofq_step_guard ORs the flags between the collectives and the optimiser -- loss <- NaN, AdamW launches update nothing -- and the
    host poll raises on all ranks at the same call (tests: eager step and the segmented graph, two ranks sharing the GPU over gloo)."""
    _spawn_two(_two_rank_fault_worker)


def test_two_processes_sharing_the_gpu_repeat_their_steps_bit_for_bit():
    """Two INDEPENDENT processes on one GPU, each repeating the same five eager training steps 150 times from the same state
    (tools/two_rank_trace.py, MODE=solo): every repetition must leave the trace of the first -- losses, every gradient, every
    parameter after every step.  Round 5's form of the attention prep's row-dot failed this in 2-5 % of the repetitions (about
    one wrong launch in 400), which is what made the two-rank test below irreproducible; 150 repetitions of each process are
    ~1500 launches of that routine per process."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, MODE="solo", REPS="150", CFGS="nodp", PROCS="2")
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "two_rank_trace.py")], cwd=root, env=env, capture_output=True, timeout=900)
    out = r.stdout.decode()
    assert r.returncode == 0, r.stderr.decode()[-3000:]
    lines = [l for l in out.splitlines() if " cfg " in l]
    assert len(lines) == 2 and all("REPRODUCIBLE" in l for l in lines), out[-3000:]


def test_two_ranks_on_one_gpu_over_gloo_run_the_data_parallel_step():
    """train.py:474, :727, :927-933 with world_size 2 for real: two processes share the GPU and exchange device tensors through
    gloo.  Gradients = mean of the ranks' local gradients; five steps of the eager DataParallel step and of the split-graph
    step (the several-rank default) give the same losses and parameters bit for bit, and the replicas stay identical."""
    import socket
    import torch.multiprocessing as mp
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    errq = ctx.SimpleQueue()
    procs = [ctx.Process(target=_two_rank_worker, args=(r, 2, port, errq)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(600)
    msgs = []
    while not errq.empty():
        msgs.append(errq.get())
    assert not msgs, "\n".join(msgs)
    assert all(p.exitcode == 0 for p in procs), [p.exitcode for p in procs]
