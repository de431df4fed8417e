#!/bin/bash
# round 6: Swin MLP / patch-merging fusions (LayerNorm + input quantiser, GELU + fc2 quantiser in fc1's epilogue): parity tests, A/B
set -u
O=gpurun_out/r06_q; mkdir -p $O
timeout 1500 python -m pytest tests/test_swin_depth_gpu.py tests/test_planes_gpu.py tests/test_planes_fullsize_gpu.py -x -q > $O/t1.txt 2>&1; echo "t1 rc=$?"; tail -3 $O/t1.txt
timeout 1500 python -m pytest tests -m gpu -x -q -k "swin or Swin" > $O/t2.txt 2>&1; echo "t2 rc=$?"; tail -3 $O/t2.txt
C4="--model swin_t --wbits 3 --abits 3 --steps 20 --warmup 5 --no-cpu-baseline"
C2="--model deit_tiny_distilled_patch16_224 --wbits 4 --abits 4 --no-qkr --batch-per-gpu 256 --steps 20 --warmup 5 --no-cpu-baseline"
run() { name=$1; shift; timeout 600 python bench.py "$@" > $O/$name.json 2> $O/$name.err; echo "$name rc=$? $(python -c "import json,sys; d=json.loads(open('$O/$name.json').read().strip().splitlines()[-1]); print(d['ms_per_step'])")"; }
for rep in 1 2; do
OFQ_NO_SWIN_MLP_FUSE=1 run c4_nofuse_$rep $C4
run c4_fuse_$rep $C4
done
run c2 $C2
run c3 --steps 20 --warmup 5 --no-cpu-baseline
