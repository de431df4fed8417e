#!/bin/bash
# round 6: Swin window partition / reverse folded into the LayerNorm passes (ofq_layernorm_lsq_fwd_perm / _bwd_perm): tests, A/B
set -u
O=gpurun_out/r06_r; mkdir -p $O
timeout 900 python -m pytest tests/test_kernels_gpu.py -x -q -k "layernorm or permutations" > $O/t0.txt 2>&1; echo "t0 rc=$?"; tail -3 $O/t0.txt
timeout 1500 python -m pytest tests/test_swin_depth_gpu.py tests/test_planes_gpu.py tests/test_planes_fullsize_gpu.py -x -q > $O/t1.txt 2>&1; echo "t1 rc=$?"; tail -3 $O/t1.txt
timeout 1500 python -m pytest tests -m gpu -x -q -k "swin or Swin" > $O/t2.txt 2>&1; echo "t2 rc=$?"; tail -3 $O/t2.txt
C4="--model swin_t --wbits 3 --abits 3 --steps 20 --warmup 5 --no-cpu-baseline"
run() { name=$1; shift; timeout 600 python bench.py "$@" > $O/$name.json 2> $O/$name.err; echo "$name rc=$? $(python -c "import json,sys; d=json.loads(open('$O/$name.json').read().strip().splitlines()[-1]); print(d['ms_per_step'])")"; }
for rep in 1 2; do
OFQ_NO_SWIN_ATTN_FUSE=1 OFQ_NO_SWIN_MLP_FUSE=1 run c4_none_$rep $C4
OFQ_NO_SWIN_ATTN_FUSE=1 run c4_mlp_$rep $C4
run c4_both_$rep $C4
done
MODEL=deit_tiny_distilled_patch16_224 BITS=4 QKR=0 BATCH=256 ONLY=ofq_absmax_f32 timeout 600 python tools/lib_call_shapes.py > $O/c2_absmax.txt 2>&1; tail -4 $O/c2_absmax.txt
run c3 --steps 20 --warmup 5 --no-cpu-baseline
