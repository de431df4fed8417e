#!/usr/bin/env python3
"""int8 forward GEMMs with the consumer quantiser in the epilogue, at the DeiT-S shapes (128 x 197 tokens):
qkx (codes only, per-(token, head) steps), fc1 (fp32 y + GELU + fc2's input codes), v (per-column steps), and the
recompute-backward of qkx.  Prints average microseconds per launch."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from ofq_amd import ops
from tools.gemm_bench import bench  # noqa

T, Bn, C, H = 198, 128, 384, 6
M = Bn * T
torch.manual_seed(0)


def operands(n, k):
    qa = torch.randint(-2, 2, (M, k), dtype=torch.int8, device="cuda")
    qw = (2 * torch.randint(-2, 2, (n, k), device="cuda") + 1).to(torch.int8)
    return qa, qw, torch.rand(T, device="cuda") * 0.05 + 0.02, torch.rand(n, device="cuda") * 0.05, torch.randn(n, device="cuda") * 0.1, \
        torch.randn(n, device="cuda") * 0.1


# qkx: N = H*C, the consumer is the per-(token, head) quantiser of q_hat k x (row mode, rowmul = H, coldiv = C), codes only
qa, qw, s, cs, bias, r = operands(H * C, C)
fq = {"s": torch.rand(T * H, device="cuda") * 0.5 + 0.5, "S": T * H, "gscale": 0.01, "b4": torch.randn(H * C, device="cuda") * 0.1,
      "lo": -2, "hi": 1, "gelu": False, "rowmul": H, "coldiv": C, "colmode": 0}
bench("i8 qkx codes-only   N=%d K=%d" % (H * C, C), lambda: ops.qgemm_i8_nt(qa, qw, None, cs, 0.25, r, s, T, 0.01, fuse=fq, store_y=False),
      2.0 * M * H * C * C)
prod = {"xcodes": qa, "wcodes": qw, "bias": None, "w_scale": cs, "w_mult": 0.25, "r": r, "act_s": s, "act_S": T, "act_gscale": 0.01}
gy = torch.randn(M, H * C, device="cuda")
bench("i8 qkx recompute-bwd N=%d K=%d" % (H * C, C), lambda: ops.qgemm_i8_lsq_bwd(gy, prod, fq), 2.0 * M * H * C * C)
del gy

# fc1: N = 4C, y stored, consumer = fc2's input quantiser after the GELU (per-token steps)
qa, qw, s, cs, bias, r = operands(4 * C, C)
f1 = {"s": torch.rand(T, device="cuda") * 0.3 + 0.2, "S": T, "gscale": 0.01, "b4": torch.randn(4 * C, device="cuda") * 0.1,
      "lo": -2, "hi": 1, "gelu": True, "rowmul": 1, "coldiv": 4 * C, "colmode": 0}
bench("i8 fc1 y+gelu+codes N=%d K=%d" % (4 * C, C), lambda: ops.qgemm_i8_nt(qa, qw, bias, cs, 0.25, r, s, T, 0.01, fuse=f1), 2.0 * M * 4 * C * C)
bench("i8 fc1 y only       N=%d K=%d" % (4 * C, C), lambda: ops.qgemm_i8_nt(qa, qw, bias, cs, 0.25, r, s, T, 0.01), 2.0 * M * 4 * C * C)

# v: N = C, consumer = per-channel quantiser (column mode)
qa, qw, s, cs, bias, r = operands(C, C)
fv = {"s": torch.rand(C, device="cuda") * 0.5 + 0.5, "S": C, "gscale": 0.01, "b4": torch.randn(C, device="cuda") * 0.1,
      "lo": -2, "hi": 1, "gelu": False, "rowmul": 1, "coldiv": C, "colmode": 1}
bench("i8 v y+codes        N=%d K=%d" % (C, C), lambda: ops.qgemm_i8_nt(qa, qw, bias, cs, 0.25, r, s, T, 0.01, fuse=fv), 2.0 * M * C * C)
# proj: plain, K = C
qa, qw, s, cs, bias, r = operands(C, C)
bench("i8 proj y only      N=%d K=%d" % (C, C), lambda: ops.qgemm_i8_nt(qa, qw, bias, cs, 0.25, r, s, T, 0.01), 2.0 * M * C * C)
# fc2: plain, K = 4 C
qa, qw, s, cs, bias, r = operands(C, 4 * C)
bench("i8 fc2 y only       N=%d K=%d" % (C, 4 * C), lambda: ops.qgemm_i8_nt(qa, qw, bias, cs, 0.25, r, s, T, 0.01), 2.0 * M * 4 * C * C)
