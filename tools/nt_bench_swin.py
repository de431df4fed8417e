#!/usr/bin/env python3
"""dX code GEMM on Swin-T shapes (many rows, few columns)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from ofq_amd import ops
from tools.gemm_bench import bench  # noqa
for (M, N, K) in [(401408, 96, 384), (401408, 96, 96), (401408, 96, 288), (100352, 192, 768), (6272, 768, 3072), (6272, 3072, 768)]:
    dy = torch.randn(M, K, device="cuda") * 1e-3
    qw = (2 * torch.randint(-4, 4, (K, N), device="cuda") + 1).to(torch.int8)
    wT = ops.codes_transpose_bf16(qw)
    ks = torch.rand(K, device="cuda")
    out = torch.empty(M, N, device="cuda")
    bench("NT dX M=%d N=%d K=%d" % (M, N, K), lambda: ops.qgemm_bf16s_nt(dy, wT, ks, 0.25, out=out), 2.0 * M * N * K)
