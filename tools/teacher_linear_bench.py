#!/usr/bin/env python3
"""The KD teacher's four linear layers of a block at 128 images (DeiT-S) on two fp16 planes each side: four plane products
against three (ofq_nt_seg.hi_only); us per launch."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from ofq_amd import ops

M = int(os.environ.get("B", 128)) * 198
ops.amax_begin(torch.device("cuda", 0))
for N, K in ((1152, 384), (384, 384), (1536, 384), (384, 1536)):
    x = torch.randn(M, K, device="cuda")
    W = torch.randn(N, K, device="cuda") * 0.05
    b = torch.randn(N, device="cuda")
    pl = ops.split_f32_f16x2(W)
    ops.absmax(x)
    res = []
    for products in (4, 3):
        for _ in range(3):
            ops.linear_f16x4(x, pl, b, products=products)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            ops.linear_f16x4(x, pl, b, products=products)
        e1.record()
        torch.cuda.synchronize()
        res.append(e0.elapsed_time(e1) * 50)
    print("N=%5d K=%5d   four %.1f us   three %.1f us" % (N, K, res[0], res[1]))
ops.amax_end()
