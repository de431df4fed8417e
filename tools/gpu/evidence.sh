#!/bin/bash
# Evidence at HEAD (run on the GPU box): kernel-trace stats of the default bench (graph replay), MFMA-utilisation and
# FETCH_SIZE / WRITE_SIZE PMC passes (eager launches: counter collection serialises kernels), traffic JSON, bench lines.
#   bash tools/gpu/evidence.sh <tag>       -> gpurun_out/evidence_<tag>/   (copy the summaries into profiles/)
set -u
TAG=${1:-head}
O=gpurun_out/evidence_$TAG; mkdir -p $O
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd /tmp
timeout 400 rocprofv3 --kernel-trace --stats -d $R/$O/kt -o kt -- python3 $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline-events > $R/$O/kt.log 2>&1; echo "kernel-trace rc=$?"
timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $R/$O/pf -o pf -- python3 $R/bench.py --steps 3 --warmup 1 --no-graph --no-cpu-baseline --no-roofline-events > $R/$O/pf.log 2>&1; echo "fetch rc=$?"
timeout 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $R/$O/pw -o pw -- python3 $R/bench.py --steps 3 --warmup 1 --no-graph --no-cpu-baseline --no-roofline-events > $R/$O/pw.log 2>&1; echo "write rc=$?"
timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $R/$O/pf6 -o pf6 -- python3 $R/bench.py --steps 6 --warmup 1 --no-graph --no-cpu-baseline --no-roofline-events > $R/$O/pf6.log 2>&1; echo "fetch6 rc=$?"
timeout 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $R/$O/pw6 -o pw6 -- python3 $R/bench.py --steps 6 --warmup 1 --no-graph --no-cpu-baseline --no-roofline-events > $R/$O/pw6.log 2>&1; echo "write6 rc=$?"
timeout 600 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_VALU -d $R/$O/pm -o pm -- python3 $R/bench.py --steps 3 --warmup 1 --no-graph --no-cpu-baseline --no-roofline-events > $R/$O/pm.log 2>&1; echo "mfma rc=$?"
cd $R
db() { find $O/$1 -name "*.db" | head -1; }
python tools/rocpd_stats.py $(db kt) 70 > $O/kernel_stats.txt; tail -1 $O/kernel_stats.txt
python tools/rocpd_gaps.py $(db kt) 0.6 > $O/gaps.txt; cat $O/gaps.txt
grep -h '"metric"' $O/kt.log | tail -1 > $O/bench_under_kernel_trace.json
python tools/pmc_summary.py $(db pf) > $O/pmc_fetch_size.txt
python tools/pmc_summary.py $(db pw) > $O/pmc_write_size.txt
python tools/pmc_mfma.py $(db pm) > $O/pmc_mfma_util.txt; head -12 $O/pmc_mfma_util.txt | cut -c1-160
python tools/make_traffic.py $(db pf) $(db pw) 4.3 $O/traffic.json $(db pf6) $(db pw6) 3 $(db pm)     # 1 warm-up + 3 timed steps + the setup_alpha forward; the 6-step passes give the steady-state step
find $O -name "*.db" -delete
timeout 300 python bench.py --steps 20 --warmup 5 > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc=$?"; cut -c1-400 $O/bench_default.json
