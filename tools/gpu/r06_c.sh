#!/bin/bash
# round 6: per-op trace of the first non-reproducible kernel (two independent contending processes)
set -u
O=gpurun_out/r06_c; mkdir -p $O
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
( time OPS=1 MODE=solo REPS=250 CFGS="nodp" timeout 900 python tools/two_rank_trace.py ) > $O/ops_solo.txt 2>&1; echo "solo rc=$?"
grep -E "cfg|rep " $O/ops_solo.txt | cut -c1-3000 | head -40
timeout 1500 python -m pytest tests/test_planes_fullsize_gpu.py -q -s > $O/planes_fullsize.txt 2>&1; echo "planes rc=$?"; grep -v "^$" $O/planes_fullsize.txt | cut -c1-180 | grep -E "passed|failed|worst|Error|assert|^family|^dx|^dw|^dq|^dxq" | head -60
