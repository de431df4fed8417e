#!/bin/bash
# round 5 experiment: what would two planes instead of three be worth? (tools/probe/bin/libofq_2plane.so = -DOFQ_2PLANE_ABLATE, wrong numbers, timing only)
set -u
O=gpurun_out/r05_ablate2; mkdir -p $O
./tools/probe/bin/f16_mfma_probe > $O/f16_probe.txt 2>&1; cat $O/f16_probe.txt
for lib in "" tools/probe/bin/libofq_2plane.so; do
  echo "== lib=${lib:-product}"
  export OFQ_HIP_LIB=${lib:+$PWD/$lib}
  [ -z "$lib" ] && unset OFQ_HIP_LIB
  CHECK=0 timeout 300 python tools/nt_sk_bench.py 2>&1 | grep -v "^$"
  SPLITS= timeout 300 python tools/tn_group_bench.py 2>&1 | tail -3
  TPWS=6 STAGGERS=0 timeout 300 python tools/dqkx_bench.py 2>&1 | tail -1
  timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline-events 2>/dev/null | cut -c1-200
done 2>&1 | tee $O/ablate.txt
