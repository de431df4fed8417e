#!/bin/bash
set -u
O=gpurun_out/r02_y; mkdir -p $O
timeout 1200 python -m pytest tests -m gpu -q -x -k "qattn or window or plain or fullsize or golden" > $O/gpu_tests.txt 2>&1; echo "tests rc=$?"
grep -E "passed|failed|Error" $O/gpu_tests.txt | tail -3
C2="--model deit_tiny_distilled_patch16_224 --wbits 4 --abits 4 --no-qkr --batch-per-gpu 256 --no-cpu-baseline --no-roofline-events --steps 20 --warmup 5"
SW="--model swin_t --wbits 3 --abits 3 --batch-per-gpu 128 --no-cpu-baseline --no-roofline-events --steps 10 --warmup 4"
run() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', d['value'], d['ms_per_step'])"; }
for i in 1 2; do
  timeout 300 python bench.py --no-cpu-baseline --no-roofline-events --steps 20 --warmup 5 2>/dev/null | run "deit-s win"
  OFQ_NO_WIN_NT=1 timeout 300 python bench.py --no-cpu-baseline --no-roofline-events --steps 20 --warmup 5 2>/dev/null | run "deit-s old"
  timeout 300 python bench.py $C2 2>/dev/null | run "c2 win"
  OFQ_NO_WIN_NT=1 timeout 300 python bench.py $C2 2>/dev/null | run "c2 old"
done
timeout 300 python bench.py $SW 2>/dev/null | run "swin win"
OFQ_NO_WIN_NT=1 timeout 300 python bench.py $SW 2>/dev/null | run "swin old"
