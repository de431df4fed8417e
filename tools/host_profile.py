#!/usr/bin/env python3
"""cProfile of the host side of the training step at a batch small enough that the GPU is never the bottleneck."""
import os, sys, cProfile, pstats
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from ofq_amd import engine
from ofq_amd.quantization.utils import KDLossSoftandHard

dev = torch.device("cuda")
B = 8
model = engine.build_student("deit_small_distilled_patch16_224", 2, 2, qk_reparam=True).to(dev)
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
pr = cProfile.Profile()
pr.enable()
for _ in range(10):
    engine.train_step(model, opt, images, target, soft, crit)
torch.cuda.synchronize()
pr.disable()
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(45)
