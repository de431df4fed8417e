#!/bin/bash
# round 6: the stem's image quantiser in patch (im2col) layout: kernel test, model tests, A/B on C3 / C2 / C4
set -u
O=gpurun_out/r06_x; mkdir -p $O
timeout 900 python -m pytest tests/test_kernels_gpu.py -x -q -k "patch_layout or lsq" > $O/t0.txt 2>&1; echo "t0 rc=$?"; tail -3 $O/t0.txt
timeout 2400 python -m pytest tests -m gpu -x -q > $O/t1.txt 2>&1; echo "t1 rc=$?"; tail -3 $O/t1.txt
C2="--model deit_tiny_distilled_patch16_224 --wbits 4 --abits 4 --no-qkr --batch-per-gpu 256"
C4="--model swin_t --wbits 3 --abits 3"
run() { name=$1; shift; timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline "$@" > $O/$name.json 2> $O/$name.err; echo "$name rc=$? $(python -c "import json,sys; d=json.loads(open('$O/$name.json').read().strip().splitlines()[-1]); print(d['ms_per_step'])")"; }
for rep in 1 2 3; do
OFQ_NO_STEM_PATCH_LAYOUT=1 run c3_old_$rep
run c3_new_$rep
done
for rep in 1 2; do
OFQ_NO_STEM_PATCH_LAYOUT=1 run c2_old_$rep $C2
run c2_new_$rep $C2
OFQ_NO_STEM_PATCH_LAYOUT=1 run c4_old_$rep $C4
run c4_new_$rep $C4
done
