#!/usr/bin/env python3
"""What a plain library fp16 GEMM (torch.matmul -> hipBLASLt, fp32 accumulation) reaches on the shapes of the two-plane input-gradient
GEMMs when the planes are given (A' = [dY_hi | dY_lo], K' = 2 K): the ceiling a split-free kernel could approach on this box at its
power cap.  Prints us and PFLOP/s of matrix-core work next to the two-plane kernel's own time on the fp32 operand."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from ofq_amd import ops

M = 128 * 198


def timeit(fn, reps=30):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps


ops.amax_begin(torch.device("cuda", 0))
for N, K in ((384, 2304), (384, 1152), (384, 384), (1536, 384), (384, 1536)):
    A = torch.randn(M, 2 * K, device="cuda", dtype=torch.float16)
    B = torch.randn(N, 2 * K, device="cuda", dtype=torch.float16)
    t_lib = timeit(lambda: torch.matmul(A, B.t()))
    dy = torch.randn(M, K, device="cuda")
    codes = torch.randint(-3, 4, (N, K), device="cuda").to(torch.float16)
    ops.absmax(dy)
    out = torch.empty(M, N, device="cuda")
    t_own = timeit(lambda: ops.qgemm_bf16s_nt(dy, codes, None, 1.0, out=out))
    fl = 2.0 * M * N * 2 * K
    print("N=%5d K=%5d  library fp16 GEMM on given planes (fp16 out) %.1f us = %.2f PF/s   own two-plane kernel on the fp32 operand %.1f us = %.2f PF/s"
          % (N, K, t_lib, fl / t_lib / 1e9, t_own, fl / t_own / 1e9))
ops.amax_end()
