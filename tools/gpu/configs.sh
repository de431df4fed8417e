#!/bin/bash
# the other BASELINE.json configurations through bench.py (parity-test cases, not bench lines) -> gpurun_out/$1/configs.txt
set -u
TAG=${1:-configs}
O=gpurun_out/$TAG; mkdir -p $O
run() { name=$1; shift; timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline-events "$@" 2>$O/$name.err | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$name', d['value'], 'img/s', d['ms_per_step'], 'ms/step', d['config']['launch'])" || { echo "$name FAILED"; tail -5 $O/$name.err; }; }
{
run c3_default
run c2_deit_t_w4a4 --model deit_tiny_distilled_patch16_224 --wbits 4 --abits 4 --no-qkr --batch-per-gpu 256
run c4_swin_t_w3a3 --model swin_t --wbits 3 --abits 3
run c5_cga --cga
run c3_with_teacher --with-teacher
run c3_force_dp --force-dp
run c3_force_dp_statsq --force-dp --sync-statsq
} | tee $O/configs.txt
