#!/bin/bash
set -u
O=gpurun_out/r02m; mkdir -p $O
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q > $O/gpu_tests.txt 2>&1; echo "gpu tests rc=$?"
tail -14 $O/gpu_tests.txt | cut -c1-300
python - <<'PY'
# input pipeline throughput: one launch over a 128-image uint8 batch
import torch, time, sys, numpy as np, random
sys.path.insert(0, '.')
from ofq_amd.data import DeviceInputPipeline, MixupParams, RandomErasingParams
x = torch.randint(0, 256, (128, 3, 224, 224), dtype=torch.uint8, device='cuda'); y = torch.randint(0, 1000, (128,), device='cuda')
pipe = DeviceInputPipeline(mixup=MixupParams(), erasing=RandomErasingParams(0.25))
for _ in range(5): pipe(x, y)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(50): pipe(x, y)
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 50
print("input pipeline (mixup/cutmix + normalise + erasing incl. noise generation): %.1f us per 128-image batch = %.0f img/s" % (dt * 1e6, 128 / dt))
PY
