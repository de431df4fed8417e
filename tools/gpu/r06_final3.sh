#!/bin/bash
# round 6, final sources: evidence.sh (kernel stats, PMC passes, traffic JSON, bench lines), the per-step launch table, the other
# configurations, kernel stats of C2 / C4, the whole GPU suite with -s, and the shared-GPU repeat traces
set -u
O=gpurun_out/r06_final6; mkdir -p $O
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R
bash tools/gpu/evidence.sh r06f > $O/evidence.log 2>&1; echo "evidence rc=$?"; tail -3 $O/evidence.log | cut -c1-300
cd /tmp
timeout 400 rocprofv3 --kernel-trace -d $R/$O/kt -o kt -- python3 $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline-events > $R/$O/kt.log 2>&1; echo "kt rc=$?"
timeout 400 rocprofv3 --kernel-trace -d $R/$O/kt4 -o kt4 -- python3 $R/bench.py --model swin_t --wbits 3 --abits 3 --steps 20 --warmup 5 --no-cpu-baseline --no-roofline-events > $R/$O/kt4.log 2>&1; echo "kt4 rc=$?"
timeout 400 rocprofv3 --kernel-trace -d $R/$O/kt2 -o kt2 -- python3 $R/bench.py --model deit_tiny_distilled_patch16_224 --wbits 4 --abits 4 --no-qkr --batch-per-gpu 256 --steps 20 --warmup 5 --no-cpu-baseline --no-roofline-events > $R/$O/kt2.log 2>&1; echo "kt2 rc=$?"
cd $R
python tools/rocpd_step.py $(find $O/kt -name "*.db" | head -1) 10 > $O/step_launches.txt 2>&1
python tools/rocpd_stats.py $(find $O/kt4 -name "*.db" | head -1) 60 > $O/kernel_stats_swin_t.txt 2>&1
python tools/rocpd_stats.py $(find $O/kt2 -name "*.db" | head -1) 60 > $O/kernel_stats_deit_t_c2.txt 2>&1
find $O -name "*.db" -delete
bash tools/gpu/configs.sh r06_final6_cfg > $O/configs.log 2>&1; cat gpurun_out/r06_final6_cfg/configs.txt
timeout 2400 python -m pytest tests -m gpu -q -s > $O/gpu_tests.txt 2>&1; echo "gpu tests rc=$?"; tail -3 $O/gpu_tests.txt
( MODE=solo REPS=300 CFGS="nodp" timeout 900 python tools/two_rank_trace.py ) > $O/trace_solo.txt 2>&1; grep -E " cfg " $O/trace_solo.txt | cut -c1-160
( MODE=ranks REPS=200 CFGS="base" timeout 900 python tools/two_rank_trace.py ) > $O/trace_ranks.txt 2>&1; grep -E " cfg " $O/trace_ranks.txt | cut -c1-160
