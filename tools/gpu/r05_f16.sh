#!/bin/bash
# round 5: the two-plane fp16 backward GEMMs -- whole GPU suite + the default bench, both plane counts
set -u
O=gpurun_out/r05_f16; mkdir -p $O
timeout 600 python -m pytest tests/test_planes_gpu.py -q -x > $O/tests_planes.txt 2>&1; echo "planes tests rc=$?"; grep -v "^$" $O/tests_planes.txt | tail -12 | cut -c1-250
timeout 1500 python -m pytest tests -m gpu -q --deselect tests/test_planes_gpu.py > $O/tests.txt 2>&1; echo "tests rc=$?"; grep -v "^$" $O/tests.txt | tail -8 | cut -c1-250
for pl in 2 3; do
  OFQ_GRAD_PLANES=$pl timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>$O/bench_$pl.err | tee $O/bench_$pl.json | cut -c1-200
done
