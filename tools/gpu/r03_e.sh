#!/bin/bash
set -u
O=gpurun_out/r03_e; mkdir -p $O
timeout 600 python -m pytest tests/test_kernels_gpu.py -q -x -k "dqkx or qattn" > $O/tests.txt 2>&1; echo "tests rc=$?"; tail -5 $O/tests.txt
for i in 1 2; do timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>$O/bench.err | cut -c1-200; done
OFQ_TN_NO_STREAM=1 timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>>$O/bench.err | cut -c1-200
