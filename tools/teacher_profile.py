#!/usr/bin/env python3
"""Where does the HIP teacher forward spend its time?  (rocprofv3 --kernel-trace --stats -- python3 tools/teacher_profile.py)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import time, torch
from ofq_amd.deit import create_model
from ofq_amd.teacher import HipTeacher
m = create_model("deit_small_distilled_patch16_224", num_classes=1000).cuda()
t = HipTeacher(m)
x = torch.randn(128, 3, 224, 224, device="cuda")
for _ in range(3):
    t(x)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(10):
    t(x)
torch.cuda.synchronize(); print("HIP teacher forward: %.2f ms" % ((time.perf_counter() - t0) * 100))
with torch.no_grad():
    for _ in range(3):
        m(x)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(10):
        m(x)
    torch.cuda.synchronize(); print("stock forward: %.2f ms" % ((time.perf_counter() - t0) * 100))
