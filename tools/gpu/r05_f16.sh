#!/bin/bash
# round 5: the two-plane fp16 backward GEMMs -- whole GPU suite + the default bench, both plane counts
set -u
O=gpurun_out/r05_f16; mkdir -p $O
timeout 2400 python -m pytest tests -m gpu -q -x > $O/tests.txt 2>&1; echo "tests rc=$?"; tail -12 $O/tests.txt | cut -c1-300
for pl in 2 3; do
  OFQ_GRAD_PLANES=$pl timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>$O/bench_$pl.err | tee $O/bench_$pl.json | cut -c1-260
done
