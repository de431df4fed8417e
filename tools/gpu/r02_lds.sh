#!/bin/bash
set -u
O=gpurun_out/r02_lds; mkdir -p $O
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd /tmp
timeout 600 rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS GRBM_GUI_ACTIVE SQ_LDS_ADDR_CONFLICT -d $R/$O/pl -o pl -- python3 $R/bench.py --steps 3 --warmup 1 --no-graph --no-cpu-baseline --no-roofline-events > $R/$O/pl.log 2>&1; echo "rc=$?"
cd $R
python tools/pmc_lds.py $(find $O/pl -name "*.db" | head -1) 16 | tee $O/pmc_lds.txt | cut -c1-150
find $O -name "*.db" -delete
