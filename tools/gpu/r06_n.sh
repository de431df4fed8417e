#!/bin/bash
# round 6: grouped dW launches for N in (128, 256) (DeiT-T's N = 192 layers, Swin stage 2): tests, then C2 / C4 / C3 timings
set -u
O=gpurun_out/r06_n; mkdir -p $O
timeout 900 python -m pytest tests/test_kernels_gpu.py tests/test_planes_gpu.py -x -q -k "tn or dw or weight_grad or group" > $O/t1.txt 2>&1; echo "t1 rc=$?"; tail -3 $O/t1.txt
timeout 1500 python -m pytest tests/test_graph_gpu.py tests/test_swin_depth_gpu.py tests/test_planes_fullsize_gpu.py -x -q > $O/t2.txt 2>&1; echo "t2 rc=$?"; tail -3 $O/t2.txt
timeout 600 python bench.py --model deit_tiny_distilled_patch16_224 --wbits 4 --abits 4 --no-qkr --batch-per-gpu 256 --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_c2.json 2> $O/bench_c2.err; echo "c2 rc=$?"; cut -c1-330 $O/bench_c2.json
timeout 600 python bench.py --model swin_t --wbits 3 --abits 3 --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_swin.json 2> $O/bench_swin.err; echo "swin rc=$?"; cut -c1-330 $O/bench_swin.json
timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_c3.json 2> $O/bench_c3.err; echo "c3 rc=$?"; cut -c1-330 $O/bench_c3.json
