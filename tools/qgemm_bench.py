#!/usr/bin/env python3
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from ofq_amd import ops
from tools.gemm_bench import bench  # noqa
M = 128 * 198
for (n, k) in [(384, 384), (1536, 384), (384, 1536), (2304, 384)]:
    qa = torch.randint(-2, 2, (M, k), dtype=torch.int8, device="cuda")
    qw = (2 * torch.randint(-2, 2, (n, k), device="cuda") + 1).to(torch.int8)
    s = torch.rand(198, device="cuda") + 0.1
    cs = torch.rand(n, device="cuda")
    bias = torch.rand(n, device="cuda")
    r = torch.rand(n, device="cuda")
    bench("i8 fwd  M=%d N=%d K=%d" % (M, n, k), lambda: ops.qgemm_i8_nt(qa, qw, bias, cs, 0.25, r, s, 198, 0.01), 2.0 * M * n * k)
    dy = torch.randn(M, n, device="cuda")
    wT = ops.codes_transpose_bf16(qw)          # [k][n]
    ks = torch.rand(n, device="cuda")
    out = torch.empty(M, k, device="cuda")
    for ns in (3, 2):
        bench("bf16x%d dX M=%d N=%d K=%d" % (ns, M, k, n), lambda: ops.qgemm_bf16s_nt(dy, wT, ks, 0.25, out=out, nsplit=ns), 2.0 * M * n * k)
