#!/usr/bin/env python3
"""Is the step host-bound?  Time the Python enqueue of N steps (no sync) against the GPU completion time, for the launch modes
of bench.py / train.py:  MODE=eager (default: one ctypes launch per kernel), graph (whole step replayed), and with DP=1 (one rank
over RCCL, the data-parallel wrapper active) eager / split (captured compute, eager collectives) / segmented (sub-graphs cut at the gradient buckets, each bucket's all-reduce behind its sub-graph: the several-rank default) /
graph (collectives captured too)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from ofq_amd import engine, parallel
from ofq_amd.quantization.utils import KDLossSoftandHard

dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
B = int(os.environ.get("B", "128"))
MODES = os.environ.get("MODE", "eager").split(",")
DP = os.environ.get("DP", "0") == "1"
if DP:
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29541")
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
for mode in MODES:
    torch.manual_seed(0)
    model = engine.build_student("deit_small_distilled_patch16_224", 2, 2, qk_reparam=True).to(dev)
    g = torch.Generator(device=dev).manual_seed(42)
    images = torch.randn(B, 3, 224, 224, device=dev, generator=g)
    target = torch.randint(0, 1000, (B,), device=dev, generator=g)
    soft = torch.randn(B, 1000, device=dev, generator=g)
    engine.setup_alpha(model, images)
    model.train()
    dp = parallel.DataParallel(model, bucket_mb=24.0, force_sync=True) if DP else None
    opt = engine.make_optimizer(model)
    crit = KDLossSoftandHard()
    if mode == "eager":
        step = lambda: engine.train_step(model, opt, images, target, soft, crit, dp=dp)      # noqa: E731
    else:
        gs = engine.GraphedTrainStep(model, opt, crit, dp=dp, warmup=2, alias_inputs=True, mode=mode if mode in ("split", "segmented") else "full")
        step = lambda: gs(images, target, soft)                                             # noqa: E731
    for _ in range(6):
        step()
    torch.cuda.synchronize()
    N = 30
    t0 = time.perf_counter()
    for _ in range(N):
        step()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print("B=%d %s%s: host enqueue %.2f ms/step, total %.2f ms/step (GPU drains %.2f ms after the last enqueue)"
          % (B, "DataParallel over RCCL (1 rank), " if DP else "", mode, 1e3 * (t1 - t0) / N, 1e3 * (t2 - t0) / N, 1e3 * (t2 - t1)),
          flush=True)
    if dp is not None:
        dp.release()
    del model, opt, dp
if DP:
    dist.destroy_process_group()
