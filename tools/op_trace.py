#!/usr/bin/env python3
"""Which torch (non-ofq) ops still launch kernels in one training step, and from where.

torch.profiler over two steps at the bench batch; prints, per aten op that launched at least one device kernel, the
count per step, the input shapes and the innermost ofq_amd / autograd frame that issued it."""
import os, sys, collections
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from torch.profiler import profile, ProfilerActivity
from ofq_amd import engine
from ofq_amd.quantization.utils import KDLossSoftandHard

dev = torch.device("cuda")
B = int(os.environ.get("B", os.environ.get("BATCH", "128")))
model = engine.build_student(os.environ.get("MODEL", "deit_small_distilled_patch16_224"), int(os.environ.get("BITS", "2")),
                             int(os.environ.get("BITS", "2")), qk_reparam=os.environ.get("QKR", "1") == "1").to(dev)
images = torch.randn(B, 3, 224, 224, device=dev)
target = torch.randint(0, 1000, (B,), device=dev)
soft = torch.randn(B, 1000, device=dev)
engine.setup_alpha(model, images)
model.train()
opt = engine.make_optimizer(model)
crit = KDLossSoftandHard()
for _ in range(3):
    engine.train_step(model, opt, images, target, soft, crit)
torch.cuda.synchronize()
STEPS = 2
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True, with_stack=True) as prof:
    for _ in range(STEPS):
        engine.train_step(model, opt, images, target, soft, crit)
    torch.cuda.synchronize()

rows = collections.Counter()
dur = collections.Counter()
for ev in prof.events():
    if not ev.name.startswith("aten::"):
        continue
    if not ev.kernels:
        continue
    # only leaf aten ops (those that own the kernels directly)
    if any(ch.kernels for ch in ev.cpu_children if ch.name.startswith("aten::")):
        continue
    frame = ""
    st = list(ev.stack or [])
    for fr in st:
        if "ofq_amd" in fr or "bench.py" in fr:
            frame = fr
            break
    if not frame:                                  # (backward ops: the autograd node's name is the best there is)
        frame = next((fr for fr in st if "Backward" in fr or "autograd" in fr), st[0] if st else "")
    frame = frame.replace(ROOT + "/", "")
    shapes = str(ev.input_shapes)[:70]
    key = (ev.name, shapes, frame[-110:])
    rows[key] += 1
    dur[key] += sum(k.duration for k in ev.kernels)
tot = 0
print("%-22s %6s %9s  %-70s %s" % ("op", "n/step", "us/step", "shapes", "frame"))
for key, n in sorted(rows.items(), key=lambda kv: -dur[kv[0]]):
    print("%-22s %6.1f %9.1f  %-70s %s" % (key[0], n / STEPS, dur[key] / STEPS, key[1], key[2]))
    tot += dur[key]
print("total torch-op kernel time per step: %.1f us, launches per step: %.1f" % (tot / STEPS, sum(rows.values()) / STEPS))
