"""GPU tests of the captured training step (engine.GraphedTrainStep): the hipGraph replay of a step must be the same
function, bit for bit, as the eager step (train.py:893-933 semantics) -- same losses, same parameters, same optimizer
state -- including the things a capture could silently freeze: the AdamW bias corrections and learning rate (which
change every step), the CGA masks (recomputed from the weights every step, cga.py:953-1013), the bucketed RCCL
all-reduce of the data-parallel wrapper, and the stem quantiser's data-dependent signedness latch (lsq.py:338-355)."""
import copy
import os

import pytest
import torch

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


def _run(base, steps, graphed, batches, cga=False, dp_factory=None, qk_reparam=True):
    from ofq_amd import engine
    from ofq_amd.quantization.utils import KDLossSoftandHard
    model = copy.deepcopy(base).train()
    dp = dp_factory(model) if dp_factory else None
    opt = engine.make_optimizer(model, lr=_lr_at(0), weight_decay=0.05)
    hooks = engine.CGAHooks(model, 3, 0.05, qk_reparam=qk_reparam) if cga else None
    loss_fn = KDLossSoftandHard()
    gs = engine.GraphedTrainStep(model, opt, loss_fn, dp=dp, cga=hooks, warmup=2) if graphed else None
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
    finally:
        dist.destroy_process_group()
