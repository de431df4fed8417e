#!/usr/bin/env python3
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from ofq_amd import ops
from tools.gemm_bench import bench
B, H, N, C = 128, 6, 198, 384
d = C // H
Np = 208
dev = "cuda"
xcodes = torch.randint(-2, 2, (B, N, C), dtype=torch.int8, device=dev)
qcodes = torch.randint(-2, 2, (B, N, H, C), dtype=torch.int8, device=dev)
vcodes = torch.randint(-2, 2, (B, N, C), dtype=torch.int8, device=dev)
pcodes = torch.zeros(B, H, N, Np, dtype=torch.int8, device=dev); pcodes[..., :N] = torch.randint(0, 4, (B, H, N, N), dtype=torch.int8, device=dev)
sx = torch.rand(N, device=dev) + .1; sq = torch.rand(N * H, device=dev) + .1; sp = torch.rand(N, device=dev) * .1 + .01; sv = torch.rand(C, device=dev) + .1
bax = torch.randn(C, device=dev) * .05; baq = torch.randn(H * C, device=dev) * .05; bav = torch.randn(C, device=dev) * .05
u = ops.rowdot_i8_multi(xcodes.view(B * N, C), baq.view(H, C)); tq = ops.rowdot_i8(qcodes.view(-1, C), bax); z = torch.mv(baq.view(H, C), bax)
fl = 2.0 * B * H * N * N
bench("rowdot_multi u", lambda: ops.rowdot_i8_multi(xcodes.view(B * N, C), baq.view(H, C)), 1)
bench("rowdot tq", lambda: ops.rowdot_i8(qcodes.view(-1, C), bax), 1)
bench("scores i8", lambda: ops.qattn_scores(xcodes, qcodes, sx, .01, sq, .01, u, tq, z, B, H, N, C, Np), fl * C)
vT = ops.codes_transpose_i8(vcodes, Np)
rp = torch.rand(B * H * N, device=dev)
bench("transpose v", lambda: ops.codes_transpose_i8(vcodes, Np), 1)
bench("pv i8", lambda: ops.qattn_pv(pcodes, vT, sp, .01, sv, .01, bav, rp, B, H, N, d, Np), fl * d)
dO = torch.randn(B, N, C, device=dev)
w = ops.rowdot_f32_seg(dO.view(B * N, C), bav, H, d)
bench("rowdot w", lambda: ops.rowdot_f32_seg(dO.view(B * N, C), bav, H, d), 1)
bench("dP bf16s", lambda: ops.qattn_dp(dO, vcodes, sv, 0.01, w, B, H, N, d, Np), fl * d)
bench("dV bf16s", lambda: ops.qattn_dv(dO, pcodes, sp, .01, B, H, N, d, Np), fl * d)
dS = torch.zeros(B, H, N, Np, device=dev); dS[..., :N] = torch.randn(B, H, N, N, device=dev)
bench("dqkx bf16s", lambda: ops.qattn_dqkx(dS, xcodes, sx, .01, bax, B, H, N, C, Np), fl * C)
bench("dxq bf16s", lambda: ops.qattn_dxq(dS, qcodes, sq, .01, B, H, N, C, Np), fl * C)
# linear ones for reference
M = B * N
for (n, k) in [(384, 384), (1536, 384), (384, 1536), (2304, 384)]:
    dy = torch.randn(M, n, device=dev); codes = torch.randint(-2, 2, (M, k), dtype=torch.int8, device=dev)
    bench("dW tn M=%d N=%d" % (n, k), lambda: ops.qgemm_bf16s_tn(dy, codes, sx, N, .01, None, bax[:k] if k <= C else torch.zeros(k, device=dev), compute_db=True), 2.0 * M * n * k)
