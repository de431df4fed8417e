#!/bin/bash
set -u
O=gpurun_out/r02s; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -q -k "swin" > $O/gpu_tests.txt 2>&1; echo "swin tests rc=$?"
tail -5 $O/gpu_tests.txt | cut -c1-300
SW="--model swin_t --wbits 3 --abits 3 --batch-per-gpu 128 --no-cpu-baseline --no-roofline-events --steps 10 --warmup 4"
for i in 1 2; do timeout 300 python bench.py $SW 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('swin', d['value'], d['ms_per_step'])"; done
