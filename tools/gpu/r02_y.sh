#!/bin/bash
set -u
C2="--model deit_tiny_distilled_patch16_224 --wbits 4 --abits 4 --no-qkr --batch-per-gpu 256 --no-cpu-baseline --no-roofline-events --steps 20 --warmup 5"
run() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', d['value'], d['ms_per_step'])"; }
for i in 1 2; do
  timeout 300 python bench.py --no-cpu-baseline --no-roofline-events --steps 20 --warmup 5 2>/dev/null | run "deit-s wink"
  OFQ_NO_WINK_TN=1 timeout 300 python bench.py --no-cpu-baseline --no-roofline-events --steps 20 --warmup 5 2>/dev/null | run "deit-s old "
  timeout 300 python bench.py $C2 2>/dev/null | run "c2 wink"
  OFQ_NO_WINK_TN=1 timeout 300 python bench.py $C2 2>/dev/null | run "c2 old "
done
