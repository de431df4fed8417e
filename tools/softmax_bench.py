#!/usr/bin/env python3
"""softmax + unsigned LSQ forward/backward on the DeiT-S attention matrix (B=128, 6 heads, 198 tokens, ld 208)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from ofq_amd import ops
from tools.gemm_bench import bench  # noqa

B, H, N, ld = 128, 6, 198, 208
rows = B * H * N
sc = torch.randn(B, H, N, ld, device="cuda")
g = torch.randn(B, H, N, ld, device="cuda") * 1e-3
s = torch.rand(N, device="cuda") * 0.1 + 0.02
n = rows * ld
prob, y, codes, rsum = ops.softmax_lsq_fwd(sc, s, rows, N, ld, N, 0.125, 3, rows, want_codes=True, need_values=False)
bench("softmax+LSQ fwd (4 r + 4 w + 1 w B/elem) TB/s:", lambda: ops.softmax_lsq_fwd(sc, s, rows, N, ld, N, 0.125, 3, rows, want_codes=True, need_values=False), 9.0 * n)
bench("softmax+LSQ bwd (8 r + 4 w B/elem)       TB/s:", lambda: ops.softmax_lsq_bwd(g, prob, s, rows, N, ld, N, 0.125, 3, rows, inplace=False, want_rowsum=True), 12.0 * n)
