#!/bin/bash
# round 5, first check: the new robustness tests + the capture stress
set -u
O=gpurun_out/r05_check1; mkdir -p $O
timeout 900 python -m pytest tests/test_kernels_gpu.py -q -x -k "stream_k or kd_loss or streaming_dx or two_segment" > $O/t_kernels.txt 2>&1; echo "kernels rc=$?"; tail -5 $O/t_kernels.txt
timeout 1200 python -m pytest tests/test_graph_gpu.py -q -x > $O/t_graph.txt 2>&1; echo "graph rc=$?"; tail -8 $O/t_graph.txt
timeout 1200 python -m pytest tests/test_training_gpu.py -q -x -s > $O/t_train.txt 2>&1; echo "train rc=$?"; tail -30 $O/t_train.txt | cut -c1-400
timeout 1200 python tools/capture_stress.py 20 > $O/capture_stress.txt 2>&1; echo "stress rc=$?"; cat $O/capture_stress.txt
STRESS_BIG=1 timeout 1500 python tools/capture_stress.py 6 > $O/capture_stress_big.txt 2>&1; echo "stress big rc=$?"; cat $O/capture_stress_big.txt
