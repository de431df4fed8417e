#!/usr/bin/env python3
"""W_qk = per-head W_q^T W_k (attention.py:190-194): forward / backward GEMM timings (DeiT-S: C=384, H=6, d=64)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from ofq_amd import ops
from ofq_amd.functional import WqkFn
from tools.gemm_bench import bench  # noqa

C, H = 384, 6
d = C // H
Wq = torch.randn(C, C, device="cuda", requires_grad=True)
Wk = torch.randn(C, C, device="cuda", requires_grad=True)
out = torch.empty(H * C, C, device="cuda")
for hint in (0, 64, 128):
    bench("Wqk fwd gemm hint=%d" % hint, lambda: ops.gemm(Wq, Wk, out, C, C, d, C, C, C, transA=True, nb0=H, sA=(d * C, 0),
                                                           sB=(d * C, 0), sC=(C * C, 0), tile_hint=hint), 2.0 * H * C * C * d)
g = torch.randn(H * C, C, device="cuda")
y = WqkFn.apply(Wq, Wk, H)
bench("WqkFn backward (2 GEMMs)", lambda: torch.autograd.grad(y, (Wq, Wk), g, retain_graph=True), 4.0 * H * C * C * d)
bench("torch einsum fwd", lambda: torch.einsum("hdc,hde->hce", Wq.view(H, d, C), Wk.view(H, d, C)), 2.0 * H * C * C * d)
dWq = torch.empty_like(Wq)
for hint in (0, 64):
    for sk in (1, 2, 4, 6):
        try:
            bench("dWq gemm hint=%d split_k=%d" % (hint, sk), lambda: ops.gemm(Wk, g, dWq, d, C, C, C, C, C, transB=True, nb0=H, sA=(d * C, 0), sB=(C * C, 0), sC=(d * C, 0), tile_hint=hint, split_k=sk), 2.0 * H * C * C * d)
        except Exception as ex:
            print("hint", hint, "split", sk, "failed:", ex)
        try:
            bench("dWk gemm hint=%d split_k=%d" % (hint, sk), lambda: ops.gemm(Wq, g, dWq, d, C, C, C, C, C, nb0=H, sA=(d * C, 0), sB=(C * C, 0), sC=(d * C, 0), tile_hint=hint, split_k=sk), 2.0 * H * C * C * d)
        except Exception as ex:
            print("hint", hint, "split", sk, "failed:", ex)
