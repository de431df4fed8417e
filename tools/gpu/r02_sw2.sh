#!/bin/bash
set -u
timeout 1500 python -m pytest tests -m gpu -q -x 2>&1 | grep -E "^[0-9]+ passed|failed|Error" | tail -3
SW="--model swin_t --wbits 3 --abits 3 --batch-per-gpu 128 --no-cpu-baseline --no-roofline-events --steps 20 --warmup 5"
run() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', d['value'], d['ms_per_step'])"; }
for i in 1 2; do
  timeout 300 python bench.py $SW 2>/dev/null | run "swin fused   "
  OFQ_NO_WINDOW_SCORES_SOFTMAX=1 timeout 300 python bench.py $SW 2>/dev/null | run "swin unfused "
  timeout 300 python bench.py $SW --no-qkr 2>/dev/null | run "swin-plain fused   "
  OFQ_NO_WINDOW_SCORES_SOFTMAX=1 timeout 300 python bench.py $SW --no-qkr 2>/dev/null | run "swin-plain unfused "
done
