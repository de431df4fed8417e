#!/bin/bash
# round 6, first box: (1) where the two-rank run stops being reproducible, (2) counters of the int8 forward / recompute kernels,
# (3) the at-size planes test, (4) the default bench line on this box.
set -u
O=gpurun_out/r06_a; mkdir -p $O
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R
( time MODE=ranks REPS=60 CFGS="base|sync|hand|bigbucket|noslot|nodefer|planes3" timeout 900 python tools/two_rank_trace.py ) > $O/trace_ranks.txt 2>&1; echo "ranks rc=$?"
grep -E "cfg|rep " $O/trace_ranks.txt | head -60
( time MODE=solo REPS=100 CFGS="nodp|base" timeout 900 python tools/two_rank_trace.py ) > $O/trace_solo.txt 2>&1; echo "solo rc=$?"
grep -E "cfg|rep " $O/trace_solo.txt | head -40
cd /tmp
rocprofv3 --list-avail > $R/$O/list_avail.txt 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU GRBM_GUI_ACTIVE -d $R/$O/p1 -o p1 -- python3 $R/tools/i8_fused_bench.py > $R/$O/p1.log 2>&1; echo "p1 rc=$?"
timeout 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_LDS SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE -d $R/$O/p2 -o p2 -- python3 $R/tools/i8_fused_bench.py > $R/$O/p2.log 2>&1; echo "p2 rc=$?"
timeout 300 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM GRBM_GUI_ACTIVE -d $R/$O/p3 -o p3 -- python3 $R/tools/i8_fused_bench.py > $R/$O/p3.log 2>&1; echo "p3 rc=$?"
timeout 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_MFMA SQ_INSTS_VALU_MFMA_MOPS_I8 SQ_WAVES SQ_INSTS_FLAT SQ_ACTIVE_INST_MISC SQ_INST_LEVEL_LDS GRBM_GUI_ACTIVE -d $R/$O/p4 -o p4 -- python3 $R/tools/i8_fused_bench.py > $R/$O/p4.log 2>&1; echo "p4 rc=$?"
cd $R
python tools/pmc_table.py $(find $O/p1 $O/p2 $O/p3 $O/p4 -name "*.db") --match qgemm_i8 > $O/pmc_i8.txt 2>&1; head -80 $O/pmc_i8.txt
python tools/i8_fused_bench.py > $O/i8_bench.txt 2>&1; cat $O/i8_bench.txt
find $O -name "*.db" -delete
timeout 1200 python -m pytest tests/test_planes_fullsize_gpu.py -x -q -s > $O/planes_fullsize.txt 2>&1; echo "planes rc=$?"; tail -5 $O/planes_fullsize.txt
timeout 600 python bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc=$?"; cut -c1-600 $O/bench_default.json
