#!/bin/bash
set -u
O=gpurun_out/r06_d; mkdir -p $O
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
( time PROCS=2 REPS=3000 timeout 900 python tools/contention_kernel_stress.py ) > $O/stress2.txt 2>&1; echo "stress2 rc=$?"
grep "^proc" $O/stress2.txt | cut -c1-700
( time PROCS=1 REPS=1500 OPS="qattn_dp_softmax_bwd,qattn_scores_softmax" timeout 600 python tools/contention_kernel_stress.py ) > $O/stress1.txt 2>&1; echo "stress1 rc=$?"
grep "^proc" $O/stress1.txt | cut -c1-700
( time OPS=1 MODE=solo REPS=250 CFGS="nodp" timeout 900 python tools/two_rank_trace.py ) > $O/ops_solo.txt 2>&1; echo "solo rc=$?"
grep -E "cfg|rep " $O/ops_solo.txt | cut -c1-900 | head -20
