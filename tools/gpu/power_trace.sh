#!/bin/bash
# socket power + shader clock (amdgpu hwmon files, every 10 ms) next to the streaming dX kernel on 99 / 198 / 256
# workgroups (two and three planes) and the default training step -> gpurun_out/$1/{power.csv, phases.txt, summary.txt}.
# The sampler is a sibling process started BEFORE the workload; it reads sysfs only and never touches the GPU runtime.
set -u
TAG=${1:-r05_power}
O=gpurun_out/$TAG; mkdir -p $O
python3 tools/power_sampler.py $O/power.csv 10 &
SP=$!
sleep 0.5
SECS=${SECS:-4} timeout 600 python3 tools/power_workload.py > $O/phases.txt 2> $O/workload.err; echo "workload rc=$?"
kill -TERM $SP; wait $SP
python3 tools/power_summary.py $O/power.csv $O/phases.txt > $O/summary.txt; cat $O/summary.txt
