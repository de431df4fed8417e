#!/usr/bin/env python3
"""Is the step host-bound?  Time the Python enqueue of N steps (no sync) against the GPU completion time."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from ofq_amd import engine
from ofq_amd.quantization.utils import KDLossSoftandHard

dev = torch.device("cuda")
B = int(os.environ.get("B", "128"))
model = engine.build_student("deit_small_distilled_patch16_224", 2, 2, qk_reparam=True).to(dev)
g = torch.Generator(device=dev).manual_seed(42)
images = torch.randn(B, 3, 224, 224, device=dev, generator=g)
target = torch.randint(0, 1000, (B,), device=dev, generator=g)
soft = torch.randn(B, 1000, device=dev, generator=g)
engine.setup_alpha(model, images)
model.train()
opt = engine.make_optimizer(model)
crit = KDLossSoftandHard()
for _ in range(5):
    engine.train_step(model, opt, images, target, soft, crit)
torch.cuda.synchronize()
N = 20
t0 = time.perf_counter()
for _ in range(N):
    engine.train_step(model, opt, images, target, soft, crit)
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print("B=%d: host enqueue %.2f ms/step, total %.2f ms/step (GPU drains %.2f ms after the last enqueue)"
      % (B, 1e3 * (t1 - t0) / N, 1e3 * (t2 - t0) / N, 1e3 * (t2 - t1)))
