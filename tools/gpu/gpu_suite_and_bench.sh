#!/bin/bash
# the whole GPU suite, the graft smoke entry, and the default bench line + the other configurations (run before a round ends)
set -u
TAG=${1:-final}
O=gpurun_out/suite_$TAG; mkdir -p $O
timeout 3000 python -m pytest tests -m gpu -q > $O/gpu_tests.txt 2>&1; echo "gpu tests rc=$?"; tail -3 $O/gpu_tests.txt
timeout 600 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.txt 2>&1; echo "smoke rc=$?"; tail -2 $O/smoke.txt
timeout 900 python bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc=$?"; cut -c1-260 $O/bench_default.json
bash tools/gpu/configs.sh suite_${TAG}_cfg > $O/configs.log 2>&1; cat gpurun_out/suite_${TAG}_cfg/configs.txt | cut -c1-100
