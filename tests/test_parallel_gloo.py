"""Multi-process (world_size 2, gloo, CPU) tests of the data-parallel layer: rank-0 broadcast of parameters and
lazily-created LSQ steps, flat-bucket gradient all-reduce launched from autograd hooks, equality with the
single-process gradient on the concatenated batch, and the StatsQ replica-consistency check."""
import os
import socket
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp
import torch.nn as nn

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


class Net(nn.Module):
    def __init__(self):
        super().__init__()
        self.a = nn.Linear(16, 32)
        self.b = nn.Linear(32, 32)
        self.c = nn.Linear(32, 4)
        self.register_buffer("signed", torch.zeros(1))
        self.frozen = nn.Parameter(torch.tensor([2.0]), requires_grad=False)

    def forward(self, x):
        return self.c(torch.relu(self.b(torch.relu(self.a(x)))))


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from ofq_amd.parallel import DataParallel, check_statsq_consistency
    torch.manual_seed(100 + rank)                       # different init on every rank ...
    net = Net()
    if rank == 0:
        net.signed.fill_(1.0)
        net.extra = nn.Parameter(torch.full((3,), 7.0))  # ... and a late-created parameter (like an LSQ `s`)
    else:
        net.extra = nn.Parameter(torch.zeros(3))
    dp = DataParallel(net, bucket_mb=0.002)             # tiny buckets -> several collectives in flight
    # rank 0 wins
    flat = torch.cat([p.detach().reshape(-1) for p in net.parameters()] + [net.signed])
    ref = flat.clone()
    dist.broadcast(ref, src=0)
    assert torch.equal(flat, ref)
    assert float(net.signed) == 1.0 and float(net.extra[0]) == 7.0
    assert len(dp.buckets) >= 3
    # one step on a rank-specific shard
    torch.manual_seed(7)
    X = torch.randn(8, 16)
    Y = torch.randn(8, 4)
    xs, ys = X[rank * 4:(rank + 1) * 4], Y[rank * 4:(rank + 1) * 4]
    dp.zero_grad()
    loss = ((dp(xs) - ys) ** 2).mean() + 0.0 * net.extra.sum()
    loss.backward()
    dp.finish_gradient_sync()
    g_dp = torch.cat([p.grad.reshape(-1) for p in net.parameters() if p.requires_grad])
    # single-process reference on the full batch
    import copy
    net2 = copy.deepcopy(net)
    for p in net2.parameters():
        p.grad = None
    (((net2(X) - Y) ** 2).mean() + 0.0 * net2.extra.sum()).backward()
    g_ref = torch.cat([p.grad.reshape(-1) for p in net2.parameters() if p.requires_grad])
    assert torch.allclose(g_dp, g_ref, rtol=1e-5, atol=1e-7), float((g_dp - g_ref).abs().max())
    # gradients are views into the flat buckets (no pack/unpack copies)
    for b in dp.buckets:
        for p in b.params:
            assert p.grad.untyped_storage().data_ptr() == b.flat.untyped_storage().data_ptr()
    # the first synchronised backward recorded the arrival order of the gradients and the buckets were rebuilt in it
    # (c before b before a: the order backward produces them), identically on both ranks
    assert dp._rebuilt
    order = [id(p) for b in dp.buckets for p in b.params]
    pos = {k: order.index(id(getattr(net, k).weight)) for k in "abc"}
    assert pos["c"] < pos["b"] < pos["a"], pos
    sizes = torch.tensor([float(len(b.params)) for b in dp.buckets] + [float(len(dp.buckets))])
    s2 = sizes.clone()
    dist.broadcast(s2, src=0)
    assert torch.equal(sizes, s2)
    # further steps work after the rebuild, and optimizer steps keep replicas identical
    opt = torch.optim.AdamW([p for p in net.parameters() if p.requires_grad], lr=1e-2)
    for _ in range(2):
        dp.zero_grad()
        assert all(p.grad is None for p in net.parameters())
        (((dp(xs) - ys) ** 2).mean() + 0.0 * net.extra.sum()).backward()
        dp.finish_gradient_sync()
        g2 = torch.cat([p.grad.reshape(-1) for p in net.parameters() if p.requires_grad])
        (g_lo, g_hi) = (g2.clone(), g2.clone())
        dist.all_reduce(g_lo, op=dist.ReduceOp.MIN)
        dist.all_reduce(g_hi, op=dist.ReduceOp.MAX)
        assert torch.equal(g_lo, g_hi)
        opt.step()
    flat = torch.cat([p.detach().reshape(-1) for p in net.parameters()])
    lo, hi = flat.clone(), flat.clone()
    dist.all_reduce(lo, op=dist.ReduceOp.MIN)
    dist.all_reduce(hi, op=dist.ReduceOp.MAX)
    assert torch.equal(lo, hi)
    # pack_only + all_reduce_packed (the two halves engine.GraphedTrainStep(mode="split") puts around its eager collectives):
    # the hooks only pack -- gradients are the LOCAL ones, already views of the buckets -- then one call reduces every bucket;
    # the result is the overlapped path's
    dp.zero_grad()
    (((dp(xs) - ys) ** 2).mean() + 0.0 * net.extra.sum()).backward()
    dp.finish_gradient_sync()
    g_sync = torch.cat([p.grad.reshape(-1) for p in net.parameters() if p.requires_grad]).clone()
    dp.pack_only = True
    dp.zero_grad()
    (((dp(xs) - ys) ** 2).mean() + 0.0 * net.extra.sum()).backward()
    dp.finish_gradient_sync()
    dp.pack_only = False
    g_local = torch.cat([p.grad.reshape(-1) for p in net.parameters() if p.requires_grad]).clone()
    for b in dp.buckets:
        for p in b.params:
            assert p.grad.untyped_storage().data_ptr() == b.flat.untyped_storage().data_ptr()
    net4 = copy.deepcopy(net)
    for p in net4.parameters():
        p.grad = None
    (((net4(xs) - ys) ** 2).mean() + 0.0 * net4.extra.sum()).backward()
    g_own = torch.cat([p.grad.reshape(-1) for p in net4.parameters() if p.requires_grad])
    assert torch.allclose(g_local, g_own, rtol=1e-6, atol=1e-8)          # nothing was reduced yet
    dp.all_reduce_packed()
    g_packed = torch.cat([p.grad.reshape(-1) for p in net.parameters() if p.requires_grad])
    assert torch.equal(g_packed, g_sync)
    # the same, bucket by bucket (engine.GraphedTrainStep(mode="segmented"): on_packed fires in the hook that packs a bucket --
    # there the step cuts its captured graph -- and all_reduce_bucket(i) / wait_collectives() reduce each bucket on its own)
    seen = []
    dp.pack_only, dp.on_packed = True, seen.append
    dp.zero_grad()
    (((dp(xs) - ys) ** 2).mean() + 0.0 * net.extra.sum()).backward()
    dp.finish_gradient_sync()
    dp.pack_only, dp.on_packed = False, None
    assert sorted(seen) == list(range(len(dp.buckets))), seen
    for i in seen:
        dp.all_reduce_bucket(i)
    dp.wait_collectives()
    g_seg = torch.cat([p.grad.reshape(-1) for p in net.parameters() if p.requires_grad])
    assert torch.equal(g_seg, g_sync)
    assert dp.check_reduced_gradients() == 0.0           # every rank holds the same reduced bytes
    # a REPLAYED step issues its collectives from the host every step and never passes through zero_grad / finish_gradient_sync
    # (they sit inside the captured graphs): the Work handles kept for drain_collectives() must not pile up (ADVICE r5:
    # len(buckets) + 1 handles leaked per step, each pinning an event and its tensors)
    for _ in range(200):
        for i in range(len(dp.buckets)):
            dp.all_reduce_bucket(i)
        dp.wait_collectives()
    assert len(dp._eager_works) <= 8 + len(dp.buckets), len(dp._eager_works)
    assert dp.drain_collectives() > 0 and dp.drain_collectives() == 0       # every eager collective so far has completed
    if rank == 1:
        dp.buckets[0].flat[0] += 1.0                       # ... and a rank whose gradients differ is caught
    with pytest.raises(RuntimeError, match="different .* gradients"):
        dp.check_reduced_gradients()
    # StatsQ statistic is a pure function of the (identical) weights: max - min over ranks must be exactly 0
    class Holder(nn.Module):
        pass
    h = Holder()
    h._s_dev = 2 * net.a.weight.detach().abs().mean(1)
    net.holder = h
    assert check_statsq_consistency(net) == 0.0
    # DDP's per-forward buffer broadcast (train.py:727) for the data-latched `signed` flag (lsq.py:338-355)
    from ofq_amd.quantization.quantizer.lsq import LsqQuantizer4img
    net3 = Net()
    del net3.signed
    net3.input_quant_fn = LsqQuantizer4img(bit=8)
    dp3 = DataParallel(net3, bucket_mb=1.0)
    qz = net3.input_quant_fn
    pos_x, neg_x = torch.rand(2, 3, 4, 4), -torch.rand(2, 3, 4, 4) - 1.0
    dp3.sync_buffers()                                   # forward 1: rank 0 sees unsigned data, rank 1 signed data
    qz._latch(neg_x if rank == 1 else pos_x)
    assert qz.latched() == (rank == 1) and not dp3._buffers_settled
    dp3.sync_buffers()                                   # forward 2: rank 0's buffer (0) overwrites rank 1's local latch
    assert not qz.latched() and float(qz.signed) == 0.0
    qz._latch(neg_x if rank == 0 else pos_x)             # now rank 0 sees negative data and latches
    assert qz.latched() == (rank == 0)
    dp3.sync_buffers()                                   # forward 3: every rank holds rank 0's 1; the broadcast has
    assert qz.latched() and float(qz.signed) == 1.0      # become the identity and is skipped from now on
    assert dp3._buffers_settled
    dp3.sync_buffers()
    # a parameter that takes no part in a step: DDP with find_unused_parameters=False raises, and so does this
    dp.zero_grad()
    ((dp(xs) - ys) ** 2).mean().backward()
    with pytest.raises(RuntimeError, match="took no part"):
        dp.finish_gradient_sync()
    q.put((rank, "ok"))
    dist.destroy_process_group()


class _SlotLinearFn(torch.autograd.Function):
    """y = x @ W^T whose backward writes dW straight into the parameter's slice of its gradient bucket
    (parallel.grad_slot), as functional.CodesLinearFn does."""

    @staticmethod
    def forward(ctx, x, W):
        ctx.save_for_backward(x, W)
        ctx.leaf = W
        return x @ W.t()

    @staticmethod
    def backward(ctx, g):
        from ofq_amd.parallel import grad_slot
        x, W = ctx.saved_tensors
        dW = g.t() @ x
        slot = grad_slot(ctx.leaf)
        if slot is not None:
            slot.copy_(dW)
            dW = slot
        return g @ W, dW


class _AllPairsFn(torch.autograd.Function):
    """out_i = q_i^T k_i for every pair, ONE node: all 2n gradients arrive together at the end of backward
    (functional.AllWqkFn)."""

    @staticmethod
    def forward(ctx, *ws):
        ctx.save_for_backward(*ws)
        return tuple(ws[2 * i].t() @ ws[2 * i + 1] for i in range(len(ws) // 2))

    @staticmethod
    def backward(ctx, *gs):
        ws = ctx.saved_tensors
        out = []
        for i, g in enumerate(gs):
            q, k = ws[2 * i], ws[2 * i + 1]
            out += [k @ g.t(), q @ g]
        return tuple(out)


class _PairNet(nn.Module):
    def __init__(self, n=6, d=8):
        super().__init__()
        self.qs = nn.ParameterList([nn.Parameter(torch.randn(4, d)) for _ in range(n)])
        self.ks = nn.ParameterList([nn.Parameter(torch.randn(4, d)) for _ in range(n)])
        self.lin = nn.ParameterList([nn.Parameter(torch.randn(d, d) * 0.3) for _ in range(n)])
        self.tied = nn.Parameter(torch.randn(d, d) * 0.3)

    def forward(self, x):
        ws = []
        for q, k in zip(self.qs, self.ks):
            ws += [q, k]
        wqk = _AllPairsFn.apply(*ws)
        for i, W in enumerate(self.lin):
            x = torch.tanh(_SlotLinearFn.apply(x, W) + x @ wqk[i])
        # one leaf feeding two nodes of the same backward: only one of them may write the bucket slice
        return _SlotLinearFn.apply(torch.tanh(_SlotLinearFn.apply(x, self.tied)), self.tied)


def _worker_slots(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import copy
    from ofq_amd.parallel import DataParallel
    torch.manual_seed(5)
    net = _PairNet()
    ref_net = copy.deepcopy(net)
    dp = DataParallel(net, bucket_mb=0.0005, sync_statsq=True)
    dp.statsq_check_every = 1
    torch.manual_seed(9)
    X = torch.randn(8, 8)
    xs = X[rank * 4:(rank + 1) * 4]
    for step in range(3):
        dp.zero_grad()
        dp(xs).pow(2).mean().backward()
        dp.finish_gradient_sync()
        for p in ref_net.parameters():
            p.grad = None
        ref_net(X).pow(2).mean().backward()
        for (n, p), (_, r) in zip(net.named_parameters(), ref_net.named_parameters()):
            assert torch.allclose(p.grad, r.grad, rtol=1e-4, atol=2e-6), (step, n, float((p.grad - r.grad).abs().max()))
        if step == 0:
            # rebuilt in arrival order: the twelve q / k weights, which arrive from one node at the very end, sit together
            # behind every linear layer's weight, each exactly once
            assert dp._rebuilt
            order = [id(p) for b in dp.buckets for p in b.params]
            assert len(order) == len(set(order)) == len(list(net.parameters()))
            late = {id(p) for p in list(net.qs) + list(net.ks)}
            first_late = min(i for i, k in enumerate(order) if k in late)
            assert all(k in late for k in order[first_late:]), "q / k gradients must be the tail of the bucket order"
    # the StatsQ-scale all-reduce is a no-op while the replicas agree ...
    class Holder(nn.Module):
        pass
    net.holder = Holder()
    net.holder._s_dev = 2 * net.tied.detach().abs().mean(1)
    dp._statsq_all_reduce()
    assert dp.check_statsq_pending() == 0.0
    # ... and raises when one rank's weights have drifted
    net.holder._s_dev = net.holder._s_dev + (1e-3 if rank == 1 else 0.0)
    dp.statsq_check_every = 10 ** 9
    dp._statsq_all_reduce()
    with pytest.raises(RuntimeError, match="StatsQ scales differ"):
        dp.check_statsq_pending()
    dp.release()
    from ofq_amd import parallel as par
    assert not any(id(p) in par._GRAD_SLOTS for p in net.parameters())
    q.put((rank, "ok"))
    dist.destroy_process_group()


def test_data_parallel_direct_write_slots_late_node_and_statsq_sync():
    """grad_slot direct writes (functional.CodesLinearFn), a node that delivers many parameters' gradients at the very end of
    backward (functional.AllWqkFn), a leaf feeding two nodes, and the optional StatsQ-scale all-reduce, world 2."""
    world = 2
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker_slots, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
    assert all(p.exitcode == 0 for p in procs), [p.exitcode for p in procs]
    got = sorted(q.get(timeout=5) for _ in range(world))
    assert got == [(0, "ok"), (1, "ok")]


def test_data_parallel_world2_gloo():
    world = 2
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
    assert all(p.exitcode == 0 for p in procs), [p.exitcode for p in procs]
    got = sorted(q.get(timeout=5) for _ in range(world))
    assert got == [(0, "ok"), (1, "ok")]
