#!/bin/bash
# round-3 baseline: per-shape dW / dX GEMM timings (default split, long-k splits) + default bench
set -u
O=gpurun_out/r03_a; mkdir -p $O
SPLITS=2,5,10 timeout 600 python tools/tn_bench.py > $O/tn_bench.txt 2>&1; echo "tn_bench rc=$?"; cat $O/tn_bench.txt
timeout 300 python bench.py --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"; cut -c1-600 $O/bench.json
