#!/usr/bin/env python3
"""Which ATen ops (not our ctypes launches) run inside one training step, with input shapes: finds stray copies / adds."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from torch.profiler import profile, ProfilerActivity
from ofq_amd import engine
from ofq_amd.quantization.utils import KDLossSoftandHard

model_name, bits, qkr, B = (sys.argv[1:] + ["deit_small_distilled_patch16_224", "2", "1", "128"])[:4]
model = engine.build_student(model_name, int(bits), int(bits), qk_reparam=bool(int(qkr))).cuda()
x = torch.randn(int(B), 3, 224, 224, device="cuda"); y = torch.randint(0, 1000, (int(B),), device="cuda"); s = torch.randn(int(B), 1000, device="cuda")
engine.setup_alpha(model, x); model.train()
opt = engine.make_optimizer(model)
lf = KDLossSoftandHard()
for _ in range(3):
    engine.train_step(model, opt, x, y, s, lf)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as prof:
    engine.train_step(model, opt, x, y, s, lf)
    torch.cuda.synchronize()
rows = [e for e in prof.key_averages(group_by_input_shape=True) if e.device_time_total > 0]
if os.environ.get("ATEN_ONLY"):
    rows = [e for e in rows if e.key.startswith("aten::")]
    print("aten:: total device us: %.1f" % sum(e.self_device_time_total for e in rows))
rows.sort(key=lambda e: -(e.count if os.environ.get("BY_COUNT") else e.device_time_total))
print("%-42s %6s %10s  %s" % ("op", "calls", "device_us", "shapes"))
if os.environ.get("ATEN_FILTER"):
    rows = [e for e in rows if os.environ["ATEN_FILTER"] in e.key]
for e in rows[:int(os.environ.get("TOP", "28"))]:
    print("%-42s %6d %10.1f  %s" % (e.key[:42], e.count, e.device_time_total, str(e.input_shapes)[:110]))
