#!/usr/bin/env python3
"""The five weight-gradient GEMMs of a DeiT-S QKR block (128 images): five launches vs one grouped launch."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from ofq_amd import ops
from tools.gemm_bench import bench  # noqa

M = 128 * 198
shapes = [(384, 1536), (1536, 384), (384, 384), (2304, 384), (384, 384)]      # (out, in) in backward order: fc2 fc1 proj qkx v
jobs = []
fl = 0.0
for (o, c) in shapes:
    dy = torch.randn(M, o, device="cuda") * 1e-3
    codes = torch.randint(-2, 2, (M, c), dtype=torch.int8, device="cuda")
    s = torch.rand(198, device="cuda") + 0.1
    baft = torch.rand(c, device="cuda")
    jobs.append({"dy2d": dy, "xcodes2d": codes, "lsq_s": s, "S": 198, "gscale": 0.01, "baft": baft,
                 "dW": torch.empty(o, c, device="cuda"), "db": torch.empty(o, device="cuda")})
    fl += 2.0 * M * o * c


def singles():
    for j in jobs:
        ops.qgemm_bf16s_tn(j["dy2d"], j["xcodes2d"], j["lsq_s"], 198, 0.01, None, j["baft"], compute_db=True, out=j["dW"])


bench("five single launches", singles, fl)
for sp in [None] + [int(x) for x in os.environ.get("SPLITS", "4,5,6,8,10,16").split(",") if x]:
    bench("grouped, split=%s" % sp, lambda: ops.qgemm_bf16s_tn_group(jobs, split=sp), fl)
bench("grouped first four (45 tiles)", lambda: ops.qgemm_bf16s_tn_group(jobs[:4]), fl - 2.0 * M * 384 * 384)
