#!/usr/bin/env python3
"""ofq_qattn_dqkx_bf16s at the DeiT-S step shape (128 images, 6 heads, 198 tokens, C = 384)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from ofq_amd import ops
from tools.gemm_bench import bench  # noqa
B, H, N, C, Np = 128, 6, 198, 384, 208
dS = torch.randn(B, H, N, Np, device="cuda") * 1e-3
xc = torch.randint(-2, 2, (B, N, C), dtype=torch.int8, device="cuda")
sx = torch.rand(N, device="cuda") + 0.1
bax = torch.rand(C, device="cuda") * 0.1
fl = 2.0 * B * H * N * N * C
for tpw in os.environ.get("TPWS", "6,1,3,12").split(","):
    os.environ["OFQ_TN_STREAM_TPW"] = tpw          # (the library's one test hook: tiles per persistent workgroup)
    bench("dqkx tpw=%s" % tpw, lambda: ops.qattn_dqkx(dS, xc, sx, 0.01, bax, B, H, N, C, Np), fl, iters=20)
