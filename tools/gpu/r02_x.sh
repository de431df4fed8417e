#!/bin/bash
set -u
O=gpurun_out/r02_x; mkdir -p $O
timeout 1200 python -m pytest tests -m gpu -q -x -k "qattn or plain or fullsize or golden or window or split" > $O/gpu_tests.txt 2>&1; echo "tests rc=$?"
grep -E "passed|failed" $O/gpu_tests.txt
for i in 1 2; do timeout 300 python bench.py --no-cpu-baseline --steps 20 --warmup 5 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('deit-s', d['value'], d['ms_per_step'])"; done
C2="--model deit_tiny_distilled_patch16_224 --wbits 4 --abits 4 --no-qkr --batch-per-gpu 256 --no-cpu-baseline --no-roofline-events --steps 20 --warmup 5"
for i in 1 2; do timeout 300 python bench.py $C2 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('c2', d['value'], d['ms_per_step'])"; done
