#!/bin/bash
set -u
O=gpurun_out/r06_g; mkdir -p $O
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
( time DUMP_OP=qattn_prep OPS=1 MODE=solo REPS=600 CFGS="nodp" timeout 900 python tools/two_rank_trace.py ) > $O/ops_solo_fix.txt 2>&1; echo "solo rc=$?"
grep -E "cfg|^    rep [0-9]+ rank|wrong value IS" $O/ops_solo_fix.txt | cut -c1-700 | head -30
timeout 900 python -m pytest tests/test_kernels_gpu.py -x -q -k "i8 or qattn or rowdot or prep" > $O/t_i8.txt 2>&1; echo "tests rc=$?"; tail -5 $O/t_i8.txt
for r in 128 64; do echo "== OFQ_I8_TILE_ROWS=$r"; OFQ_I8_TILE_ROWS=$r python tools/i8_fused_bench.py 2>&1 | grep "i8 "; done | tee $O/i8_rows.txt
