#!/usr/bin/env python3
"""Stream-K dX GEMM against the number of workgroups, interleaved with the one-tile-per-workgroup kernel (rounds of 10 launches
each, median over the rounds: boxes drift by several per cent within a process).  With a -DNTSK_CLOCK_PROBE build of the
library in OFQ_HIP_LIB the shader clock a mid-grid workgroup saw is printed as well: does filling the idle CUs pay, or does the
chip give the gain back in clock?"""
import os, sys, statistics
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from ofq_amd import ops

M = int(os.environ.get("IMGS", "128")) * 198
WGS = [int(x) for x in os.environ.get("WGS", "128,198,256").split(",")]
ROUNDS = int(os.environ.get("ROUNDS", "5"))
probe = bool(os.environ.get("OFQ_HIP_LIB"))
torch.manual_seed(0)


def timed(fn, iters=10):
    s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e3


def clock(g, dev):
    ws = ops._sk_workspace(dev)
    fl = ws[:32768].view(torch.int32)
    cyc, tick = int(fl[4098].item()), int(fl[4099].item())
    return cyc / max(tick, 1) / 10.0


def operands(o, c):
    dy = torch.randn(M, o, device="cuda") * 1e-3
    qw = (2 * torch.randint(-2, 2, (o, c), device="cuda") + 1).to(torch.int8)
    return dy, ops.codes_transpose_bf16(qw), torch.rand(o, device="cuda") + 0.5


def run(name, variants, flops):
    for fn in variants.values():
        fn(); fn()
    t = {k: [] for k in variants}
    for _ in range(ROUNDS):
        for k, fn in variants.items():
            t[k].append(timed(fn))
    for k in variants:
        med = statistics.median(t[k])
        extra = ""
        if probe and k.startswith("G="):
            variants[k](); torch.cuda.synchronize()
            extra = "  %.2f GHz" % clock(int(k[2:]), torch.device("cuda", 0))
        print("%-28s %-10s %8.1f us (min %6.1f)  %6.1f TF/s%s" % (name, k, med, min(t[k]), flops / med / 1e6, extra), flush=True)


for (o, c) in [(384, 384), (1536, 384), (384, 1536), (2304, 384)]:
    dy, wT, ks = operands(o, c)
    out = torch.empty(M, c, device="cuda")
    v = {"classic": lambda: ops.qgemm_bf16s_nt(dy, wT, ks, 0.25, out=out, sk=False)}
    for g in WGS:
        v["G=%d" % g] = (lambda g=g: ops.qgemm_bf16s_nt_sk([(dy, wT, ks, 0.25)], out, wgs=g))
    run("N=%d K=%d" % (c, o), v, 2.0 * M * o * c)

dv, wv, kv = operands(384, 384)
dq, wq, kq = operands(2304, 384)
out = torch.empty(M, 384, device="cuda")


def two():
    ops.qgemm_bf16s_nt(dv, wv, kv, 0.25, out=out, sk=False)
    ops.qgemm_bf16s_nt(dq, wq, kq, 0.25, out=out, accumulate=True, sk=False)


v = {"classic x2": two}
for g in WGS:
    v["G=%d" % g] = (lambda g=g: ops.qgemm_bf16s_nt_sk([(dv, wv, kv, 0.25), (dq, wq, kq, 0.25)], out, wgs=g))
run("v + W_qk (K = 384 + 2304)", v, 2.0 * M * 384 * 2688)
