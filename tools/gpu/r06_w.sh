#!/bin/bash
set -u
O=gpurun_out/r06_w; mkdir -p $O
timeout 1800 python -m pytest tests -m gpu -x -q -k "plain or deit_tiny or golden or depth12 or graph_replay or w4a4" > $O/t1.txt 2>&1; echo "t1 rc=$?"; tail -3 $O/t1.txt
C2="--model deit_tiny_distilled_patch16_224 --wbits 4 --abits 4 --no-qkr --batch-per-gpu 256 --steps 20 --warmup 5 --no-cpu-baseline"
run() { name=$1; shift; timeout 600 python bench.py "$@" > $O/$name.json 2> $O/$name.err; echo "$name rc=$? $(python -c "import json,sys; d=json.loads(open('$O/$name.json').read().strip().splitlines()[-1]); print(d['ms_per_step'])")"; }
for rep in 1 2 3; do
OFQ_NO_PLAIN_PREP=1 run c2_old_$rep $C2
run c2_new_$rep $C2
done
run c3 --steps 20 --warmup 5 --no-cpu-baseline
