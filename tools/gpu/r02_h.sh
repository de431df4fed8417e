#!/bin/bash
set -u
O=gpurun_out/r02h; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -q -k "fullsize or train_cli or cga_hooks" > $O/gpu_tests.txt 2>&1; echo "gpu tests rc=$?"
tail -8 $O/gpu_tests.txt | cut -c1-300
SW="--model swin_t --wbits 3 --abits 3 --batch-per-gpu 128 --no-cpu-baseline"
timeout 300 python bench.py --steps 10 --warmup 4 $SW > $O/swin.json 2> $O/swin.err; echo "swin rc=$?"
python - <<PY
import json
try:
    d=json.loads(open("$O/swin.json").read().strip().splitlines()[-1]); print("swin", d["value"], d["ms_per_step"], d["config"].get("launch")); print("   ", (d.get("roofline") or {}).get("all_classes_ms_per_step"))
except Exception as e: print("swin failed", e); print(open("$O/swin.err").read()[-1500:])
PY
cd /tmp
timeout 400 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$O/prof -o p -- python3 $GRAFT_REPO_ROOT/bench.py --steps 6 --warmup 3 --no-roofline-events $SW > $GRAFT_REPO_ROOT/$O/prof.log 2>&1; echo "rocprof rc=$?"
cd $GRAFT_REPO_ROOT
db=$(find $O/prof -name "*.db" | head -1)
python tools/rocpd_stats.py $db 60 > $O/kernel_stats_swin.txt; tail -1 $O/kernel_stats_swin.txt
python tools/rocpd_stats.py $db 400 grid | grep -E "qgemm_i8_nt_kernel<[12]>|qgemm_bf16s_nt_kernel<3, true>|qgemm_bf16s_tn_kernel|qgemm_bf16s_nn_kernel|tn_wide|nn_wide" > $O/kernel_stats_swin_grid.txt
find $O/prof -name "*.db" -delete
