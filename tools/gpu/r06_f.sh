#!/bin/bash
set -u
O=gpurun_out/r06_f; mkdir -p $O
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
( time DETAIL=1 PROCS=2 REPS=30000 OPS="qattn_prep" timeout 900 python tools/contention_kernel_stress.py ) > $O/stress_prep.txt 2>&1; echo "stress rc=$?"
grep "^proc" $O/stress_prep.txt | cut -c1-400 | head -40
( time PROCS=1 REPS=30000 OPS="qattn_prep" timeout 900 python tools/contention_kernel_stress.py ) > $O/stress_prep1.txt 2>&1; echo "stress1 rc=$?"
grep "^proc" $O/stress_prep1.txt | cut -c1-400 | head
