#!/bin/bash
set -u
O=gpurun_out/r03_f; mkdir -p $O
timeout 600 python -m pytest tests/test_kernels_gpu.py -q -x -k "dp_softmax or qattn" > $O/tests.txt 2>&1; echo "tests rc=$?"; tail -15 $O/tests.txt
timeout 900 python -m pytest tests/test_modules_gpu.py tests/test_prod_gpu.py -q -x > $O/mods.txt 2>&1; echo "mods rc=$?"; tail -5 $O/mods.txt
for i in 1 2; do timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>$O/bench.err | cut -c1-200; done
OFQ_NO_DP_SOFTMAX_FUSE=1 timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>>$O/bench.err | cut -c1-200
