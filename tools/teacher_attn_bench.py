#!/usr/bin/env python3
"""The KD teacher's attention at 128 images (DeiT-S): ofq_attn_f32_fwd against the three launches it replaces (strided
fp32-MFMA scores, softmax kernel, strided fp32-MFMA P.V); us per call."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from ofq_amd import ops
from ofq_amd.functional import pad4

B, H, N, d = int(os.environ.get("B", 128)), 6, 198, 64
C = H * d
qkv = torch.randn(B * N, 3 * C, device="cuda")
ones = torch.ones(N, device="cuda")
Np = pad4(N)


def three():
    S = torch.empty((B, H, N, Np), dtype=torch.float32, device="cuda")
    ops.gemm(qkv, qkv, S, N, N, d, 3 * C, 3 * C, Np, transB=True, nb0=B, nb1=H, sA=(N * 3 * C, d), sB=(N * 3 * C, d),
             sC=(H * N * Np, N * Np), offB=C)
    P, _ = ops.softmax_lsq_fwd(S, ones, B * H * N, N, Np, N, d ** -0.5, 1, 1, need_values=False)
    O = torch.empty((B * N, C), dtype=torch.float32, device="cuda")
    ops.gemm(P, qkv, O, N, d, N, Np, 3 * C, C, nb0=B, nb1=H, sA=(H * N * Np, N * Np), sB=(N * 3 * C, d), sC=(N * C, d), offB=2 * C)
    return O


def one():
    return ops.attn_f32_fwd(qkv, B, H, N, d, d ** -0.5)


def timeit(fn, reps=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps


a, b = three(), one()
print("max |one - three| / max|three| = %.3g" % ((a - b).abs().max() / a.abs().max()).item())
for _ in range(3):
    print("three launches %.1f us   one launch %.1f us" % (timeit(three), timeit(one)))
