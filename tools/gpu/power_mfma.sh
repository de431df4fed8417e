#!/bin/bash
# Socket power + shader clock next to kernels that issue nothing but MFMAs: the matrix-core rate the part sustains under its power
# cap -> gpurun_out/$1/{power.csv, phases.txt, summary.txt}
set -u
TAG=${1:-r05_power_mfma}
O=gpurun_out/$TAG; mkdir -p $O
[ -f tools/probe/bin/libmfma_power.so ] || /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -shared -fPIC tools/probe/mfma_power_probe.hip -o tools/probe/bin/libmfma_power.so
python3 tools/power_sampler.py $O/power.csv 10 &
SP=$!
sleep 0.5
SECS=${SECS:-3} timeout 900 python3 tools/power_mfma_workload.py > $O/phases.txt 2> $O/workload.err; echo "workload rc=$?"
kill -TERM $SP; wait $SP
python3 tools/power_summary.py $O/power.csv $O/phases.txt > $O/summary.txt; cat $O/summary.txt
