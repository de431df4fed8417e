#!/bin/bash
# attention-kernel tests + kernel-trace of the default bench -> gpurun_out/$1
set -u
TAG=${1:-attn}
O=gpurun_out/$TAG; mkdir -p $O
timeout 900 python -m pytest tests/test_kernels_gpu.py tests/test_modules_gpu.py tests/test_prod_gpu.py -q -x > $O/tests.txt 2>&1; echo "tests rc=$?"; tail -4 $O/tests.txt
bash tools/gpu/kernel_trace.sh $TAG | grep "TOTAL\|metric\|softmax\|stream" | cut -c1-200
