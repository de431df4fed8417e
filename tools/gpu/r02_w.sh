#!/bin/bash
set -u
O=gpurun_out/r02_w; mkdir -p $O
export TMPDIR=/tmp
timeout 1200 python -m pytest tests -m gpu -q -x -k "swin or lsq or permute or layernorm or Swin" > $O/gpu_tests.txt 2>&1; echo "tests rc=$?"
tail -5 $O/gpu_tests.txt | cut -c1-300
SW="--model swin_t --wbits 3 --abits 3 --batch-per-gpu 128 --no-cpu-baseline --no-roofline-events --steps 10 --warmup 4"
for i in 1 2; do timeout 300 python bench.py $SW 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('swin', d['value'], d['ms_per_step'])"; done
R=$GRAFT_REPO_ROOT
cd /tmp
timeout 400 rocprofv3 --kernel-trace --stats -d $R/$O/kt -o kt -- python3 $R/bench.py $SW > $R/$O/kt.log 2>&1; echo "rc=$?"
cd $R
python tools/rocpd_stats.py $(find $O/kt -name "*.db" | head -1) 60 > $O/kernel_stats_swin.txt
python tools/rocpd_stats.py $(find $O/kt -name "*.db" | head -1) 120 grid > $O/kernel_stats_swin_grid.txt
find $O -name "*.db" -delete
