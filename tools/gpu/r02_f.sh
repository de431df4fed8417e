#!/bin/bash
# round 2, call F: plain attention on codes: parity (goldens + full size), C2 bench A/B
set -u
O=gpurun_out/r02f; mkdir -p $O
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q > $O/gpu_tests.txt 2>&1; echo "gpu tests rc=$?"
tail -25 $O/gpu_tests.txt | cut -c1-300
C2="--model deit_tiny_distilled_patch16_224 --wbits 4 --abits 4 --no-qkr --batch-per-gpu 256 --no-cpu-baseline"
run() { # name, env, args
  env $2 timeout 300 python bench.py --steps 20 --warmup 5 $3 > $O/$1.json 2> $O/$1.err
  python - <<PY
import json
try:
    d=json.loads(open("$O/$1.json").read().strip().splitlines()[-1]); print("$1", d["value"], d["ms_per_step"], d["config"].get("launch")); print("   ", (d.get("roofline") or {}).get("all_classes_ms_per_step"))
except Exception as e: print("$1 failed", e); print(open("$O/$1.err").read()[-1500:])
PY
}
run c2_codes "A=1" "$C2"
run c2_fp32 "OFQ_NO_PLAIN_ATTN_CODES=1" "$C2"
run c2_codes_eager "A=1" "$C2 --no-graph"
run c3 "A=1" "--no-cpu-baseline"
cd /tmp
timeout 400 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$O/prof -o p -- python3 $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 3 --no-roofline-events $C2 > $GRAFT_REPO_ROOT/$O/prof.log 2>&1; echo "rocprof rc=$?"
cd $GRAFT_REPO_ROOT
db=$(find $O/prof -name "*.db" | head -1)
python tools/rocpd_stats.py $db 50 > $O/kernel_stats_c2.txt; tail -1 $O/kernel_stats_c2.txt
find $O/prof -name "*.db" -delete
