#!/usr/bin/env python3
"""Micro-benchmark of the LSQ forward/backward kernels on the DeiT-S step shapes (B=128): achieved HBM GB/s."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from ofq_amd import ops
from tools.gemm_bench import bench  # noqa

B, S = 128, 197
for (inner, prologue, name) in [(384, 0, "per-token act 384"), (1536, 1, "fc2 input 1536 (GELU)"), (1152, 0, "qkx 1152")]:
    x = torch.randn(B * S, inner, device="cuda")
    gy = torch.randn(B * S, inner, device="cuda") * 1e-3
    s = torch.rand(S, device="cuda") * 0.5 + 0.2
    b4 = torch.randn(inner, device="cuda") * 0.1
    baft = torch.randn(inner, device="cuda") * 0.1
    g = ops.LsqGeom(B, S, inner, inner, 0, -2, 1, B * inner, prologue=prologue)
    n = B * S * inner
    def fwd():
        ops.lsq_fwd(x, s, b4, baft, g, want_codes=True, need_values=False)
    def bwd():
        ops.lsq_bwd(gy, x, s, b4, g)
    # bench() prints "TF/s" = arg / time / 1e12: pass bytes so the column reads TB/s
    bench("LSQ fwd %-24s (5 B/elem) TB/s:" % name, fwd, 5.0 * n)
    bench("LSQ bwd %-24s (12 B/elem) TB/s:" % name, bwd, 12.0 * n)
