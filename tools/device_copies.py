#!/usr/bin/env python3
"""Which host call sites issue the small device copies / fills of one training step (torch.profiler with stacks)."""
import os, sys, collections
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from torch.profiler import profile, ProfilerActivity
from ofq_amd import engine
from ofq_amd.quantization.utils import KDLossSoftandHard

dev = torch.device("cuda")
B = 32
model = engine.build_student("deit_small_distilled_patch16_224", 2, 2, qk_reparam=True).to(dev)
images = torch.randn(B, 3, 224, 224, device=dev)
target = torch.randint(0, 1000, (B,), device=dev)
soft = torch.randn(B, 1000, device=dev)
engine.setup_alpha(model, images)
model.train()
opt = engine.make_optimizer(model, lr=5e-4, weight_decay=0.05)
crit = KDLossSoftandHard()
def step():
    return engine.train_step(model, opt, images, target, soft, crit, dp=None, cga=None)
for _ in range(2):
    step()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU], with_stack=True, record_shapes=True) as prof:
    step()
    torch.cuda.synchronize()
want = sys.argv[1:] or ["aten::copy_", "aten::fill_", "aten::zero_", "aten::zeros", "aten::zeros_like", "aten::add", "aten::add_",
                        "aten::contiguous", "aten::clone", "aten::sum", "aten::mul", "aten::div", "aten::mean", "aten::neg"]
cnt = collections.Counter()
for ev in prof.events():
    if ev.name in want:
        st = [f for f in (ev.stack or []) if "ofq_amd" in f or "bench" in f or "torch/optim" in f][:2]
        shp = str(ev.input_shapes)[:60]
        cnt[(ev.name, shp, " <- ".join(s.split("/")[-1] for s in st))] += 1
for (k, v) in cnt.most_common(60):
    print(v, k)

print("---- all aten ops by count")
allc = collections.Counter(ev.name for ev in prof.events() if ev.name.startswith("aten::"))
for (k, v) in allc.most_common(40):
    print(v, k)
