#!/bin/bash
# round 6, on frozen kernel sources: evidence.sh (kernel stats, PMC passes, traffic JSON, bench lines), the whole GPU suite with -s (the
# planes-at-size tables), and the shared-GPU repeat traces (two independent processes x 300, two ranks x 200)
set -u
O=gpurun_out/r06_final2; mkdir -p $O
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
bash tools/gpu/evidence.sh r06b > $O/evidence.log 2>&1; tail -12 $O/evidence.log | cut -c1-200
timeout 1800 python -m pytest tests -q -m gpu -s > $O/gpu_tests.txt 2>&1; echo "gpu tests rc=$?"; grep -E "passed|failed" $O/gpu_tests.txt | tail -3
( MODE=solo REPS=300 CFGS="nodp" timeout 600 python tools/two_rank_trace.py ) > $O/trace_solo.txt 2>&1; grep " cfg " $O/trace_solo.txt | cut -c1-200
( MODE=ranks REPS=200 CFGS="base" timeout 600 python tools/two_rank_trace.py ) > $O/trace_ranks.txt 2>&1; grep " cfg " $O/trace_ranks.txt | cut -c1-200
