#!/usr/bin/env python3
"""Stream-K input-gradient GEMM (ofq_qgemm_bf16s_nt_sk) against the one-tile-per-workgroup kernel on the DeiT-S step shapes:
values (fp64 reference on sampled rows, and the classic kernel), launch-to-launch determinism, time.  CHECK=0 skips the checks."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from ofq_amd import ops
from tools.gemm_bench import bench  # noqa

B = int(os.environ.get("IMGS", "128"))
M = B * 198
check = os.environ.get("CHECK", "1") != "0"
torch.manual_seed(0)


def operands(o, c):
    dy = torch.randn(M, o, device="cuda") * 1e-3 * torch.logspace(-4, 0, M, device="cuda")[torch.randperm(M, device="cuda")].unsqueeze(1)   # rows over 4 decades
    if os.environ.get("DYN"):       # element-wise dynamic range of DYN decades (what real gradient tensors look like)
        dy = torch.randn(M, o, device="cuda") * 1e-3 * torch.pow(10.0, -float(os.environ["DYN"]) * torch.rand(M, o, device="cuda"))
    qw = (2 * torch.randint(-2, 2, (o, c), device="cuda") + 1).to(torch.int8)
    wT = ops.codes_transpose_16(qw)       # fp16 codes (two-plane form) unless OFQ_GRAD_PLANES=3
    ks = torch.rand(o, device="cuda") + 0.5
    return dy, qw, wT, ks


def ref_rows(dy, qw, ks, alpha, rows):
    return alpha * ((dy[rows].double() * ks.double()) @ qw.double())


for (o, c) in [(384, 384), (1536, 384), (384, 1536), (2304, 384)]:
    dy, qw, wT, ks = operands(o, c)
    out0 = torch.empty(M, c, device="cuda")
    out1 = torch.empty(M, c, device="cuda")
    if check:
        ops.qgemm_bf16s_nt(dy, wT, ks, 0.25, out=out0, sk=False)
        ops.qgemm_bf16s_nt(dy, wT, ks, 0.25, out=out1, sk=True)
        rows = torch.arange(0, M, 97, device="cuda")
        ref = ref_rows(dy, qw, ks, 0.25, rows)
        e0 = ((out0[rows].double() - ref).abs().max() / ref.abs().max()).item()
        e1 = ((out1[rows].double() - ref).abs().max() / ref.abs().max()).item()
        d = ((out1 - out0).abs().max() / out0.abs().max()).item()
        same = True
        for _ in range(5):
            out2 = torch.empty(M, c, device="cuda").fill_(float("nan"))
            ops.qgemm_bf16s_nt(dy, wT, ks, 0.25, out=out2, sk=True)
            same = same and torch.equal(out1, out2)
        # accumulate form
        base = torch.randn(M, c, device="cuda")
        acc0 = base.clone(); acc1 = base.clone()
        ops.qgemm_bf16s_nt(dy, wT, ks, 0.25, out=acc0, accumulate=True, sk=False)
        ops.qgemm_bf16s_nt(dy, wT, ks, 0.25, out=acc1, accumulate=True, sk=True)
        da = ((acc1 - acc0).abs().max() / acc0.abs().max()).item()
        print("N=%d K=%d: classic vs fp64 %.2e, stream-K vs fp64 %.2e, stream-K vs classic %.2e, accumulate %.2e, deterministic %s, err word %d"
              % (c, o, e0, e1, d, da, same, ops.nt_sk_error(dy.device)), flush=True)
    bench("NT dX classic  M=%d N=%d K=%d" % (M, c, o), lambda: ops.qgemm_bf16s_nt(dy, wT, ks, 0.25, out=out0, sk=False), 2.0 * M * o * c)
    bench("NT dX stream-K M=%d N=%d K=%d" % (M, c, o), lambda: ops.qgemm_bf16s_nt(dy, wT, ks, 0.25, out=out1, sk=True), 2.0 * M * o * c)

# two K-segments: v (K = 384) + W_qk (K = 2304) into one x_hat gradient
dy_v, qw_v, wT_v, ks_v = operands(384, 384)
dy_q, qw_q, wT_q, ks_q = operands(2304, 384)
out0 = torch.empty(M, 384, device="cuda")
out1 = torch.empty(M, 384, device="cuda")


def two_launches():
    ops.qgemm_bf16s_nt(dy_v, wT_v, ks_v, 0.25, out=out0, sk=False)
    ops.qgemm_bf16s_nt(dy_q, wT_q, ks_q, 0.25, out=out0, accumulate=True, sk=False)


def two_sk():
    ops.qgemm_bf16s_nt(dy_v, wT_v, ks_v, 0.25, out=out0, sk=True)
    ops.qgemm_bf16s_nt(dy_q, wT_q, ks_q, 0.25, out=out0, accumulate=True, sk=True)


def concat():
    ops.qgemm_bf16s_nt_sk([(dy_v, wT_v, ks_v, 0.25), (dy_q, wT_q, ks_q, 0.25)], out1)


if check:
    two_launches(); concat()
    rows = torch.arange(0, M, 97, device="cuda")
    ref = ref_rows(dy_v, qw_v, ks_v, 0.25, rows) + ref_rows(dy_q, qw_q, ks_q, 0.25, rows)
    e0 = ((out0[rows].double() - ref).abs().max() / ref.abs().max()).item()
    e1 = ((out1[rows].double() - ref).abs().max() / ref.abs().max()).item()
    keep = out1.clone()
    same = True
    for _ in range(5):
        out1.fill_(float("nan")); concat()
        same = same and torch.equal(out1, keep)
    print("v + W_qk: two launches vs fp64 %.2e, concatenated stream-K vs fp64 %.2e, deterministic %s, err word %d"
          % (e0, e1, same, ops.nt_sk_error(out1.device)), flush=True)
fl = 2.0 * M * 384 * (384 + 2304)
bench("v + W_qk dX: two classic launches", two_launches, fl)
bench("v + W_qk dX: two stream-K launches", two_sk, fl)
bench("v + W_qk dX: one concatenated stream-K launch", concat, fl)
