#!/bin/bash
set -u
O=gpurun_out/r02k; mkdir -p $O
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_kernels_gpu.py -x -q -k "fused_scores or attention_prep" > $O/kernel_tests.txt 2>&1; echo "kernel tests rc=$?"
tail -15 $O/kernel_tests.txt | cut -c1-300
timeout 1500 python -m pytest tests -m gpu -q > $O/gpu_tests.txt 2>&1; echo "gpu tests rc=$?"
tail -12 $O/gpu_tests.txt | cut -c1-300
B="--steps 20 --warmup 5 --no-cpu-baseline --no-roofline-events"
run() { env $2 timeout 300 python bench.py $B $3 > $O/$1.json 2> $O/$1.err
  python - <<PY
import json
try:
    d=json.loads(open("$O/$1.json").read().strip().splitlines()[-1]); print("$1", d["value"], d["ms_per_step"], d["config"].get("launch"), d["config"]["loss"])
except Exception as e: print("$1 failed", e); print(open("$O/$1.err").read()[-1500:])
PY
}
run prep "A=1" ""
run noprep "OFQ_NO_ATTN_PREP=1" ""
run prep2 "A=1" ""
run noprep2 "OFQ_NO_ATTN_PREP=1" ""
C2="--model deit_tiny_distilled_patch16_224 --wbits 4 --abits 4 --no-qkr --batch-per-gpu 256"
run c2_fused "A=1" "$C2"
run c2_unfused "OFQ_NO_SCORES_SOFTMAX_FUSE=1" "$C2"
cd /tmp
timeout 400 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$O/prof -o p -- python3 $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-roofline-events > $GRAFT_REPO_ROOT/$O/prof.log 2>&1; echo "rocprof rc=$?"
cd $GRAFT_REPO_ROOT
db=$(find $O/prof -name "*.db" | head -1)
python tools/rocpd_stats.py $db 30 > $O/kernel_stats.txt; grep -E "scores_softmax|softmax_lsq|i8_nt_kernel<1" $O/kernel_stats.txt | cut -c1-160; tail -1 $O/kernel_stats.txt
find $O/prof -name "*.db" -delete
