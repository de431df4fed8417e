#!/bin/bash
# round 6, second box: full-tensor detail of the first differing gradients (two independent processes contending / two ranks),
# packed-fp32 rate probe, at-size planes test
set -u
O=gpurun_out/r06_b; mkdir -p $O
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R
( time FULL=1 MODE=solo REPS=150 CFGS="nodp" timeout 600 python tools/two_rank_trace.py ) > $O/trace_solo_full.txt 2>&1; echo "solo rc=$?"
grep -E "cfg|rep |\('" $O/trace_solo_full.txt | cut -c1-260 | head -120
( time FULL=1 MODE=ranks REPS=80 CFGS="base|bigbucket" timeout 600 python tools/two_rank_trace.py ) > $O/trace_ranks_full.txt 2>&1; echo "ranks rc=$?"
grep -E "cfg|rep |\('" $O/trace_ranks_full.txt | cut -c1-260 | head -150
./tools/probe/bin/pk_rate_probe > $O/pk_rate.txt 2>&1; cat $O/pk_rate.txt
timeout 1500 python -m pytest tests/test_planes_fullsize_gpu.py -x -q -s > $O/planes_fullsize.txt 2>&1; echo "planes rc=$?"; grep -v "^$" $O/planes_fullsize.txt | tail -150
