#!/usr/bin/env python3
"""Which piece of the plain-attention code path deviates at full size?  Compares S (codes) with the fp64 product of the
dequantised operands, for several batch sizes / head counts / bit widths."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from ofq_amd import ops
from ofq_amd.functional import pad16

torch.manual_seed(0)
for (B, N, C, H, bits) in [(2, 198, 192, 3, 4), (64, 198, 192, 3, 4), (256, 198, 192, 3, 4), (256, 198, 192, 3, 2),
                           (128, 198, 384, 6, 4), (256, 198, 384, 6, 2)]:
    d = C // H
    lo, hi = -(2 ** (bits - 1)), 2 ** (bits - 1) - 1
    qc = torch.randint(lo, hi + 1, (B, N, C), dtype=torch.int8, device="cuda")
    kc = torch.randint(lo, hi + 1, (B, N, C), dtype=torch.int8, device="cuda")
    sq = torch.rand(N, device="cuda") * 0.3 + 0.1
    sk = torch.rand(N, device="cuda") * 0.3 + 0.1
    bq = torch.randn(C, device="cuda") * 0.05
    bk = torch.randn(C, device="cuda") * 0.05
    Np = pad16(N)
    eye = torch.eye(H, device="cuda").repeat_interleave(d, dim=1)
    u = ops.rowdot_i8_multi(qc.view(B * N, C), eye * bk)
    tq = ops.rowdot_i8_multi(kc.view(B * N, C), eye * bq)
    z = (bq * bk).view(H, d).sum(1)
    S = ops.qattn_scores_plain(qc, kc, sq, 0.0, sk, 0.0, u, tq, z, B, H, N, d, Np)
    qh = (qc.double() * sq.double()[None, :, None] + bq.double()).view(B, N, H, d).permute(0, 2, 1, 3)
    kh = (kc.double() * sk.double()[None, :, None] + bk.double()).view(B, N, H, d).permute(0, 2, 1, 3)
    ref = qh @ kh.transpose(-1, -2)
    u_ref = (qc.double().view(B, N, H, d) * bk.double().view(H, d)).sum(-1).view(B * N, H)
    err = (S[..., :N].double() - ref).abs()
    print("B=%d C=%d H=%d bits=%d: S max err %.3e (max |S| %.2f)  u err %.3e   worst (b,h) = %s" % (
        B, C, H, bits, float(err.max()), float(ref.abs().max()), float((u.double() - u_ref).abs().max()),
        tuple(int(v) for v in torch.unravel_index(err.amax((2, 3)).argmax(), (B, H)))))
    dS = torch.randn(B, H, N, Np, device="cuda")
    dq = ops.qattn_dq_plain(dS, kc, sk, 0.0, B, H, N, d, Np)
    dk = ops.qattn_dk_plain(dS, qc, sq, 0.0, bq, B, H, N, d, Np)
    dq_ref = (dS[..., :N].double() @ (kc.double() * sk.double()[None, :, None]).view(B, N, H, d).permute(0, 2, 1, 3))
    dk_ref = dS[..., :N].double().transpose(-1, -2) @ qh
    print("     dq err %.3e  dk err %.3e" % (float((dq.view(B, N, H, d).permute(0, 2, 1, 3).double() - dq_ref).abs().max()),
                                             float((dk.view(B, N, H, d).permute(0, 2, 1, 3).double() - dk_ref).abs().max())))
