#!/bin/bash
set -u
O=gpurun_out/r06_e; mkdir -p $O
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
( time DUMP_OP=qattn_prep OPS=1 MODE=solo REPS=400 CFGS="nodp" timeout 900 python tools/two_rank_trace.py ) > $O/ops_solo4.txt 2>&1; echo "solo rc=$?"
grep -E "^    rep [0-9]+ rank|wrong value IS" $O/ops_solo4.txt | cut -c1-300 | head -80
