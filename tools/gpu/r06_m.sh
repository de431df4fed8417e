#!/bin/bash
# round 6: step guard (stream-K error seen by every rank) + wide dW kernels for short step vectors (Swin MLP): tests, then timings
set -u
O=gpurun_out/r06_m; mkdir -p $O
timeout 900 python -m pytest tests/test_kernels_gpu.py -x -q -k "stream_k or step_guard or engine_step or tn" > $O/t1.txt 2>&1; echo "t1 rc=$?"; tail -3 $O/t1.txt
timeout 900 python -m pytest tests/test_planes_gpu.py -x -q > $O/t2.txt 2>&1; echo "t2 rc=$?"; tail -3 $O/t2.txt
timeout 1500 python -m pytest tests/test_graph_gpu.py -x -q > $O/t3.txt 2>&1; echo "t3 rc=$?"; tail -5 $O/t3.txt
timeout 1500 python -m pytest tests/test_swin_depth_gpu.py tests/test_planes_fullsize_gpu.py -x -q -s > $O/t4.txt 2>&1; echo "t4 rc=$?"; tail -3 $O/t4.txt
timeout 600 python bench.py --model swin_t --wbits 3 --abits 3 --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_swin.json 2> $O/bench_swin.err; echo "swin rc=$?"; cut -c1-330 $O/bench_swin.json
timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_c3.json 2> $O/bench_c3.err; echo "c3 rc=$?"; cut -c1-330 $O/bench_c3.json
timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --force-dp > $O/bench_dp.json 2> $O/bench_dp.err; echo "dp rc=$?"; cut -c1-330 $O/bench_dp.json
