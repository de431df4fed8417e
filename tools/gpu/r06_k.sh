#!/bin/bash
set -u
O=gpurun_out/r06_k; mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_kernels_gpu.py tests/test_fullsize_gpu.py -x -q -k "128x384 or i8" > $O/t.txt 2>&1; echo "tests rc=$?"; tail -3 $O/t.txt | cut -c1-250
echo "== old kernel"; OFQ_I8_L384=0 python tools/i8_fused_bench.py 2>&1 | grep -E "i8 (v|proj|fc2)"
echo "== shipped"; python tools/i8_fused_bench.py 2>&1 | grep -E "i8 (v|proj|fc2)"
for v in 0 1 0 1; do OFQ_I8_L384=$v python bench.py --steps 50 --warmup 10 --no-cpu-baseline --no-roofline-events --no-recipe-line | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('OFQ_I8_L384=$v', d['ms_per_step'], 'ms/step', d['value'], 'img/s')"; done
