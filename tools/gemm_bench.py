#!/usr/bin/env python3
"""Micro-benchmark of ofq_gemm_f32 on the shapes of the DeiT-S W2A2 QKR step (B=128/GPU) + a 4096^3 reference."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from ofq_amd import ops

def bench(name, fn, flops, iters=10):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record(); torch.cuda.synchronize()
    ms = s.elapsed_time(e) / iters
    print("%-46s %9.1f us  %7.1f TF/s" % (name, ms * 1e3, flops / ms / 1e9), flush=True)

def rnd(*shape):
    return torch.randn(*shape, device="cuda")

def main():
    hints = [int(h) for h in os.environ.get("HINTS", "0,128,64").split(",")]
    B, N, C, H = 128, 198, 384, 6
    M = B * N
    d = C // H
    Np = 200
    for (n, k) in [(384, 384), (1536, 384), (384, 1536), (2304, 384)]:
        x, W, y = rnd(M, k), rnd(n, k), torch.empty(M, n, device="cuda")
        for h in hints:
            bench("NT  M=%d N=%d K=%d hint=%d" % (M, n, k, h), lambda: ops.gemm(x, W, y, M, n, k, k, k, n, transB=True, tile_hint=h), 2.0 * M * n * k)
        dy, dx = rnd(M, n), torch.empty(M, k, device="cuda")
        for h in hints:
            bench("NN  M=%d N=%d K=%d hint=%d" % (M, k, n, h), lambda: ops.gemm(dy, W, dx, M, k, n, n, k, k, tile_hint=h), 2.0 * M * n * k)
        dW = torch.empty(n, k, device="cuda")
        sp = ops._pick_split(n, k, M)
        for h in hints:
            bench("TN  M=%d N=%d K=%d split=%d hint=%d" % (n, k, M, sp, h), lambda: ops.gemm(dy, x, dW, n, k, M, n, k, k, transA=True, split_k=sp, tile_hint=h), 2.0 * M * n * k)
    xq, qkx, S = rnd(B, N, C), rnd(B, N, H, C), torch.empty(B, H, N, Np, device="cuda")
    for h in hints:
        bench("S=xq.qkx^T  768x(198,198,384) hint=%d" % h, lambda: ops.gemm(xq, qkx, S, N, N, C, C, H * C, Np, transB=True, nb0=B, nb1=H, sA=(N * C, 0), sB=(N * H * C, C), sC=(H * N * Np, N * Np), tile_hint=h), 2.0 * B * H * N * N * C)
    P = torch.zeros(B, H, N, Np, device="cuda"); P[..., :N] = torch.rand(B, H, N, N, device="cuda")
    v, O = rnd(B, N, C), torch.empty(B, N, C, device="cuda")
    bench("O=P.V  768x(198,64,198)", lambda: ops.gemm(P, v, O, N, d, N, Np, C, C, nb0=B, nb1=H, sA=(H * N * Np, N * Np), sB=(N * C, d), sC=(N * C, d)), 2.0 * B * H * N * N * d)
    dxq = torch.empty(B, N, C, device="cuda")
    for h in hints:
        bench("dxq=sum_h dS.qkx 128x(198,384,6x198) hint=%d" % h, lambda: ops.gemm(P, qkx, dxq, N, C, N, Np, H * C, C, nb0=B, sA=(H * N * Np, 0), sB=(N * H * C, 0), sC=(N * C, 0), nkb=H, sAk=N * Np, sBk=C, tile_hint=h), 2.0 * B * H * N * N * C)
    dq = torch.empty(B, N, H, C, device="cuda")
    for h in hints:
        bench("dqkx=dS^T.xq 768x(198,384,198) hint=%d" % h, lambda: ops.gemm(P, xq, dq, N, C, N, Np, C, H * C, transA=True, nb0=B, nb1=H, sA=(H * N * Np, N * Np), sB=(N * C, 0), sC=(N * H * C, C), tile_hint=h), 2.0 * B * H * N * N * C)
    A4, B4, C4 = rnd(4096, 4096), rnd(4096, 4096), torch.empty(4096, 4096, device="cuda")
    bench("NT 4096^3", lambda: ops.gemm(A4, B4, C4, 4096, 4096, 4096, 4096, 4096, 4096, transB=True), 2.0 * 4096 ** 3)
    bench("NN 4096^3", lambda: ops.gemm(A4, B4, C4, 4096, 4096, 4096, 4096, 4096, 4096), 2.0 * 4096 ** 3)
    bench("TN 4096^3", lambda: ops.gemm(A4, B4, C4, 4096, 4096, 4096, 4096, 4096, 4096, transA=True), 2.0 * 4096 ** 3)


if __name__ == '__main__':
    main()
