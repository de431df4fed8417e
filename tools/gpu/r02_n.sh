#!/bin/bash
set -u
O=gpurun_out/r02n; mkdir -p $O
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q > $O/gpu_tests.txt 2>&1; echo "gpu tests rc=$?"
tail -4 $O/gpu_tests.txt | cut -c1-300
timeout 300 python tools/op_profile.py deit_tiny_distilled_patch16_224 4 0 256 2>&1 | grep -v amdgpu.ids > $O/ops_c2.txt; head -32 $O/ops_c2.txt
timeout 300 python tools/op_profile.py deit_small_distilled_patch16_224 2 1 128 2>&1 | grep -v amdgpu.ids > $O/ops_c3.txt; head -24 $O/ops_c3.txt
C2="--model deit_tiny_distilled_patch16_224 --wbits 4 --abits 4 --no-qkr --batch-per-gpu 256 --no-cpu-baseline --no-roofline-events"
timeout 300 python bench.py --steps 20 --warmup 5 $C2 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('c2', d['value'], d['ms_per_step'])"
