#!/usr/bin/env python3
"""Micro-benchmark of the weight-gradient (TN) and input-gradient (NT) code GEMMs on the DeiT-S step shapes (B=128)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from ofq_amd import ops
from tools.gemm_bench import bench  # noqa

M = 128 * 198
which = os.environ.get("WHICH", "tn,nt").split(",")
for (o, c) in [(2304, 384), (384, 384), (1536, 384), (384, 1536)]:
    dy = torch.randn(M, o, device="cuda") * 1e-3
    codes = torch.randint(-2, 2, (M, c), dtype=torch.int8, device="cuda")
    s = torch.rand(198, device="cuda") + 0.1
    baft = torch.rand(c, device="cuda")
    if "tn" in which:
        for split in [None] + [int(x) for x in os.environ.get("SPLITS", "").split(",") if x]:
            bench("TN dW o=%d c=%d split=%s" % (o, c, split),
                  lambda: ops.qgemm_bf16s_tn(dy, codes, s, 198, 0.01, None, baft, split=split, compute_db=True), 2.0 * M * o * c)
    if "nt" in which:
        qw = (2 * torch.randint(-2, 2, (o, c), device="cuda") + 1).to(torch.int8)
        wT = ops.codes_transpose_bf16(qw)
        ks = torch.rand(o, device="cuda")
        out = torch.empty(M, c, device="cuda")
        bench("NT dX M=%d N=%d K=%d" % (M, c, o), lambda: ops.qgemm_bf16s_nt(dy, wT, ks, 0.25, out=out), 2.0 * M * o * c)
