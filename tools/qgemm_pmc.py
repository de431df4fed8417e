#!/usr/bin/env python3
"""A few launches of each bf16-split kernel at DeiT-S shapes (for rocprofv3 --pmc runs)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from ofq_amd import ops
M, n, k = 128 * 198, 1536, 384
dev = "cuda"
dy = torch.randn(M, n, device=dev)
qw = (2 * torch.randint(-2, 2, (n, k), device=dev) + 1).to(torch.int8)
wT = ops.codes_transpose_bf16(qw)
ks = torch.rand(n, device=dev)
out = torch.empty(M, k, device=dev)
codes = torch.randint(-2, 2, (M, k), dtype=torch.int8, device=dev)
sx = torch.rand(198, device=dev) + .1
baft = torch.randn(k, device=dev) * .05
for _ in range(3):
    ops.qgemm_bf16s_nt(dy, wT, ks, 0.25, out=out, nsplit=3)
    ops.qgemm_bf16s_tn(dy, codes, sx, 198, .01, None, baft, compute_db=True)
torch.cuda.synchronize()
