#!/usr/bin/env python3
"""LSQ forward/backward on Swin-T stage-1 shapes (B=128: 8192 windows x 49 tokens, C=96, 3 heads)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from ofq_amd import ops
from tools.gemm_bench import bench  # noqa

for (outer, S, inner, k, name) in [(8192, 49, 96, 1, "x  [8192,49,96]"), (8192, 147, 96, 3, "qkx [8192,49*3,96] k=3"),
                                   (8192, 49, 384, 1, "mlp hidden [8192,49,384]"), (2048, 49, 192, 1, "stage2 x")]:
    x = torch.randn(outer * S, inner, device="cuda")
    gy = torch.randn(outer * S, inner, device="cuda") * 1e-3
    s = torch.rand(S, device="cuda") * 0.5 + 0.2
    b4 = torch.randn(k * inner, device="cuda") * 0.1
    baft = torch.randn(k * inner, device="cuda") * 0.1
    g = ops.LsqGeom(outer, S, inner, k * inner, 0, -4, 3, outer * inner)
    n = outer * S * inner
    bench("LSQ fwd %-28s TB/s (5 B/elem):" % name, lambda: ops.lsq_fwd(x, s, b4, baft, g, want_codes=True, need_values=False), 5.0 * n)
    bench("LSQ bwd %-28s TB/s (12 B/elem):" % name, lambda: ops.lsq_bwd(gy, x, s, b4, g), 12.0 * n)
