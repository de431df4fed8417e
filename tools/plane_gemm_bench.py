#!/usr/bin/env python3
"""ofq_gemm_bf16x3x3_nt (the fp32 KD teacher's linear layers) on the DeiT-S shapes at 128 images."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from ofq_amd import ops
from tools.gemm_bench import bench  # noqa
M = 128 * 198
for (n, k) in [(1152, 384), (384, 384), (1536, 384), (384, 1536)]:
    x = torch.randn(M, k, device="cuda")
    W = torch.randn(n, k, device="cuda") * 0.05
    b = torch.randn(n, device="cuda")
    pl = ops.split_f32_bf16x3(W)
    for prod in (9, 6):
        bench("plane GEMM M=%d N=%d K=%d products=%d" % (M, n, k, prod), lambda: ops.gemm_bf16x3x3_nt(x, pl, b, products=prod),
              2.0 * M * n * k)
