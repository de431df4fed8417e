#!/usr/bin/env python3
"""LayerNorm forward/backward: csrc/layernorm.hip vs the stock PyTorch kernels on the DeiT-S token matrix (B=128)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from ofq_amd import ops
from tools.gemm_bench import bench  # noqa

R, C = 128 * 198, 384
x = torch.randn(R, C, device="cuda")
res = torch.randn(R, C, device="cuda")
dy = torch.randn(R, C, device="cuda")
g = torch.rand(C, device="cuda") + 0.5
b = torch.rand(C, device="cuda")
n = R * C
y, xs, mean, rstd = ops.layernorm_fwd(x, g, b, 1e-6, res2d=res)
bench("HIP LN fwd            ( 8 B/elem) TB/s:", lambda: ops.layernorm_fwd(x, g, b, 1e-6), 8.0 * n)
bench("HIP add+LN fwd        (16 B/elem) TB/s:", lambda: ops.layernorm_fwd(x, g, b, 1e-6, res2d=res), 16.0 * n)
bench("HIP LN bwd            (12 B/elem) TB/s:", lambda: ops.layernorm_bwd(dy, x, mean, rstd, g), 12.0 * n)
bench("HIP LN bwd + dres     (16 B/elem) TB/s:", lambda: ops.layernorm_bwd(dy, x, mean, rstd, g, dres2d=res), 16.0 * n)
xt = x.clone().requires_grad_(True)
gt, bt = g.clone().requires_grad_(True), b.clone().requires_grad_(True)
bench("torch LN fwd          ( 8 B/elem) TB/s:", lambda: torch.nn.functional.layer_norm(xt, (C,), gt, bt, 1e-6), 8.0 * n)
yt = torch.nn.functional.layer_norm(xt, (C,), gt, bt, 1e-6)
bench("torch LN bwd          (12 B/elem) TB/s:", lambda: torch.autograd.grad(yt, (xt, gt, bt), dy, retain_graph=True), 12.0 * n)
