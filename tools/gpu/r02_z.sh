#!/bin/bash
# every bench configuration once (sanity + numbers for the docs)
set -u
run() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', d['value'], d['ms_per_step'])"; }
F="--no-cpu-baseline --no-roofline-events --steps 20 --warmup 5"
timeout 300 python bench.py $F 2>/dev/null | run "C3 deit-s W2A2 QKR 128"
timeout 300 python bench.py $F --cga 2>/dev/null | run "C5 deit-s CGA"
timeout 300 python bench.py $F --with-teacher 2>/dev/null | run "deit-s + teacher bf16x9"
timeout 300 python bench.py $F --force-dp 2>/dev/null | run "deit-s DP 1 rank (RCCL)"
timeout 300 python bench.py $F --no-graph 2>/dev/null | run "deit-s eager"
timeout 300 python bench.py $F --model deit_tiny_distilled_patch16_224 --wbits 4 --abits 4 --no-qkr --batch-per-gpu 256 2>/dev/null | run "C2 deit-t W4A4 256"
timeout 300 python bench.py $F --model swin_t --wbits 3 --abits 3 2>/dev/null | run "C4 swin-t W3A3 128"
timeout 300 python bench.py $F --model swin_t --wbits 3 --abits 3 --no-qkr 2>/dev/null | run "swin-t W3A3 plain 128"
