#!/bin/bash
# round 6 evidence at the final sources: GPU suite, kernel trace + PMC passes + traffic (evidence.sh), the other configurations,
# kernel stats of C2 / C4, the full-size soak, the default bench line
set -u
O=gpurun_out/r06_final; mkdir -p $O
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
timeout 1800 python -m pytest tests -q -m gpu -s > $O/gpu_tests.txt 2>&1; echo "gpu tests rc=$?"; grep -E "passed|failed" $O/gpu_tests.txt | tail -3
bash tools/gpu/evidence.sh r06 > $O/evidence.log 2>&1; tail -25 $O/evidence.log | cut -c1-200
bash tools/gpu/configs.sh r06_configs > $O/configs.log 2>&1; cat gpurun_out/r06_configs/configs.txt
BENCH_ARGS="--model deit_tiny_distilled_patch16_224 --wbits 4 --abits 4 --no-qkr --batch-per-gpu 256" bash tools/gpu/kernel_trace.sh r06_kt_c2 > $O/kt_c2.log 2>&1; tail -3 $O/kt_c2.log | cut -c1-200
BENCH_ARGS="--model swin_t --wbits 3 --abits 3" bash tools/gpu/kernel_trace.sh r06_kt_c4 > $O/kt_c4.log 2>&1; tail -3 $O/kt_c4.log | cut -c1-200
( PCHK=0 STEPS=400 MODE=eager timeout 600 python tools/step_soak_determinism.py ) > $O/soak_eager.txt 2>&1; grep -E "^proc" $O/soak_eager.txt | cut -c1-200
( PCHK=0 STEPS=1500 MODE=graph timeout 600 python tools/step_soak_determinism.py ) > $O/soak_graph.txt 2>&1; grep -E "^proc" $O/soak_graph.txt | cut -c1-200
timeout 600 python bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc=$?"; cut -c1-300 $O/bench_default.json
