#!/usr/bin/env python3
"""Where does the data-parallel wrapper's overhead go?  One rank, RCCL backend, forced hooks / collectives."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29555")
os.environ.setdefault("RANK", "0"); os.environ.setdefault("WORLD_SIZE", "1")
import torch
import torch.distributed as dist
from ofq_amd import engine, parallel
from ofq_amd.quantization.utils import KDLossSoftandHard

dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
dist.init_process_group(backend="nccl", device_id=dev)
B = 128
model = engine.build_student("deit_small_distilled_patch16_224", 2, 2, qk_reparam=True).to(dev)
g = torch.Generator(device=dev).manual_seed(42)
images = torch.randn(B, 3, 224, 224, device=dev, generator=g)
target = torch.randint(0, 1000, (B,), device=dev, generator=g)
soft = torch.randn(B, 1000, device=dev, generator=g)
engine.setup_alpha(model, images)
model.train()
crit = KDLossSoftandHard()


def run(dp, label, n=20):
    opt = engine.make_optimizer(model)
    for _ in range(5):
        engine.train_step(model, opt, images, target, soft, crit, dp=dp)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        engine.train_step(model, opt, images, target, soft, crit, dp=dp)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print("%-44s enqueue %.2f ms/step   total %.2f ms/step" % (label, 1e3 * (t1 - t0) / n, 1e3 * (t2 - t0) / n), flush=True)


run(None, "no wrapper")
dp = parallel.DataParallel(model, bucket_mb=24.0, force_sync=True)
run(dp, "DataParallel, 4 x 24 MB buckets, hooks")
for h in dp._hooks:
    h.remove()
for p in model.parameters():
    p.grad = None
dp2 = parallel.DataParallel(model, bucket_mb=24.0, force_sync=True)
for h in dp2._hooks:
    h.remove()
dp2._hooks = []
run(dp2, "DataParallel, buckets launched after backward")
dist.destroy_process_group()
