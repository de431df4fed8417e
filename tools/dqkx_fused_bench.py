#!/usr/bin/env python3
"""The qkx chain of one DeiT-S block's backward, 128 images: ofq_qattn_dqkx_bf16s (stream kernel, two fp16 planes) followed by
ofq_qgemm_i8_lsq_bwd, against the fused ofq_qattn_dqkx_lsq_bwd.  Prints us per launch (HIP events, cold-ish operands: the
buffers of 4 rotating problem instances are larger than the L2 + MALL)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from ofq_amd import ops

B, H, N, C = int(os.environ.get("B", 128)), 6, 198, 384
ldS = 208
M, Nout = B * N, H * C
g = torch.Generator(device="cuda").manual_seed(1)
inst = []
for k in range(3):
    xc = torch.randint(-2, 2, (M, C), dtype=torch.int8, device="cuda", generator=g)
    wc = (2 * torch.randint(-2, 2, (Nout, C), device="cuda", generator=g) + 1).to(torch.int8)
    dS = torch.randn(B, H, N, ldS, device="cuda", generator=g) * 1e-3
    inst.append((xc, wc, dS))
bias = torch.randn(Nout, device="cuda", generator=g) * 0.1
cs = torch.rand(Nout, device="cuda", generator=g) * 0.05 + 0.01
r = torch.randn(Nout, device="cuda", generator=g) * 0.3
sx = torch.rand(N, device="cuda", generator=g) * 0.3 + 0.05
bax = torch.randn(C, device="cuda", generator=g) * 0.2
qs = torch.rand(N * H, device="cuda", generator=g) * 0.3 + 0.1
qb4 = torch.randn(Nout, device="cuda", generator=g) * 0.05
q = dict(s=qs, S=N * H, gscale=0.021, b4=qb4, lo=-2, hi=1, gelu=0, rowmul=H, coldiv=C, colmode=0)


def prod(xc, wc):
    return {"xcodes": xc, "wcodes": wc, "bias": bias, "w_scale": cs, "w_mult": 0.25, "r": r, "act_s": sx, "act_S": N, "act_gscale": 0.013}


def pair(xc, wc, dS):
    gy = ops.qattn_dqkx(dS, xc, sx, 0.013, bax, B, H, N, C, ldS, planes=2)
    return ops.qgemm_i8_lsq_bwd(gy.view(M, Nout), prod(xc, wc), q)


def fused(xc, wc, dS):
    return ops.qattn_dqkx_lsq_bwd(dS, prod(xc, wc), q, bax, B, H, N, C, ldS, planes=2)


def timeit(fn, reps=12):
    for i in range(3):
        fn(*inst[i % 3])
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(reps):
        fn(*inst[i % 3])
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps


ops.amax_begin(torch.device("cuda", 0))
for xc, wc, dS in inst:
    ops.absmax(dS.view(-1, ldS)[:, :N])
a = pair(*inst[0]); b = fused(*inst[0])
print("equal:", [bool(torch.equal(x, y)) for x, y in zip(a, b)])
for rnd in range(3):
    print("pair  %.1f us   fused %.1f us" % (timeit(pair), timeit(fused)))
ops.amax_end()
