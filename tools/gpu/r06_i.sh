#!/bin/bash
set -u
O=gpurun_out/r06_i; mkdir -p $O
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
rocm-smi --showuse --showmemuse 2>/dev/null | head -20
( PCHK=0 STEPS=400 MODE=eager timeout 600 python tools/step_soak_determinism.py ) > $O/soak_eager_b.txt 2>&1; grep -E "^proc|step " $O/soak_eager_b.txt | cut -c1-300
( PCHK=0 STEPS=1500 MODE=graph timeout 600 python tools/step_soak_determinism.py ) > $O/soak_graph_b.txt 2>&1; grep -E "^proc|step " $O/soak_graph_b.txt | cut -c1-300
( PCHK=0 PROCS=2 STEPS=300 MODE=eager timeout 900 python tools/step_soak_determinism.py ) > $O/soak_eager_2procs.txt 2>&1; grep -E "^proc|step " $O/soak_eager_2procs.txt | cut -c1-300
( PCHK=0 PROCS=2 STEPS=600 MODE=graph timeout 900 python tools/step_soak_determinism.py ) > $O/soak_graph_2procs.txt 2>&1; grep -E "^proc|step " $O/soak_graph_2procs.txt | cut -c1-300
