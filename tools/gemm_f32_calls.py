#!/usr/bin/env python3
"""Every ofq_gemm_f32 launch of one training step with its shape and HIP-event time (which fp32-MFMA GEMMs are left?)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from ofq_amd import engine, ops
from ofq_amd.quantization.utils import KDLossSoftandHard

model_name, bits, qkr, B = (sys.argv[1:] + ["deit_small_distilled_patch16_224", "2", "1", "128"])[:4]
model = engine.build_student(model_name, int(bits), int(bits), qk_reparam=bool(int(qkr))).cuda()
x = torch.randn(int(B), 3, 224, 224, device="cuda"); y = torch.randint(0, 1000, (int(B),), device="cuda"); s = torch.randn(int(B), 1000, device="cuda")
engine.setup_alpha(model, x); model.train()
opt = engine.make_optimizer(model)
lf = KDLossSoftandHard()
for _ in range(3):
    engine.train_step(model, opt, x, y, s, lf)
torch.cuda.synchronize()
calls = []
orig = ops.gemm
def logged(A, B_, Cout, M, N, K, lda, ldb, ldc, **kw):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); r = orig(A, B_, Cout, M, N, K, lda, ldb, ldc, **kw); e1.record()
    calls.append((M, N, K, kw.get("transA", False), kw.get("transB", False), kw.get("nb0", 1), kw.get("nb1", 1), kw.get("split_k", 1), e0, e1))
    return r
ops.gemm = logged
import ofq_amd.functional as F_
for mod in list(sys.modules.values()):
    if mod is not None and getattr(mod, "__name__", "").startswith("ofq_amd") and getattr(mod, "gemm", None) is orig:
        mod.gemm = logged
engine.train_step(model, opt, x, y, s, lf)
torch.cuda.synchronize()
tot = 0.0
print("%8s %6s %6s  tA tB  nb0 nb1 split   us     GF")
for M, N, K, ta, tb, nb0, nb1, sk, e0, e1 in calls:
    us = e0.elapsed_time(e1) * 1e3; tot += us
    print("%8d %6d %6d  %d  %d  %4d %3d %4d %7.1f %7.2f" % (M, N, K, ta, tb, nb0, nb1, sk, us, 2e-9 * M * N * K * nb0 * nb1))
print("total %.1f us in %d launches" % (tot, len(calls)))
