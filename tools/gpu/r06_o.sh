#!/bin/bash
# round 6: same-box A/B of the dW routing changes (wide kernels for S < 32; grouped launches for N in (128, 256)) on C4 and C2
set -u
O=gpurun_out/r06_o; mkdir -p $O
C4="--model swin_t --wbits 3 --abits 3 --steps 20 --warmup 5 --no-cpu-baseline"
C2="--model deit_tiny_distilled_patch16_224 --wbits 4 --abits 4 --no-qkr --batch-per-gpu 256 --steps 20 --warmup 5 --no-cpu-baseline"
run() { name=$1; shift; timeout 600 python bench.py "$@" > $O/$name.json 2> $O/$name.err; echo "$name rc=$? $(python -c "import json,sys; d=json.loads(open('$O/$name.json').read().strip().splitlines()[-1]); print(d['ms_per_step'])")"; }
for rep in 1 2; do
OFQ_TN_NARROW_SMALL_S=1 OFQ_TN_GROUP_MIN_N=256 OFQ_TN_GROUP_MIN_S=32 run c4_old_$rep $C4
OFQ_TN_GROUP_MIN_N=256 OFQ_TN_GROUP_MIN_S=32 run c4_wideS_$rep $C4
OFQ_TN_GROUP_MIN_N=256 run c4_wideS_groupS_$rep $C4
run c4_new_$rep $C4
OFQ_TN_GROUP_MIN_N=256 run c2_old_$rep $C2
run c2_new_$rep $C2
done
