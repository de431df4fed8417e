#!/bin/bash
set -u
O=gpurun_out/r02j; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -q -k "swin or qattn or attention" > $O/gpu_tests.txt 2>&1; echo "gpu tests rc=$?"
tail -5 $O/gpu_tests.txt | cut -c1-300
SW="--model swin_t --wbits 3 --abits 3 --batch-per-gpu 128 --no-cpu-baseline --no-roofline-events --steps 10 --warmup 4"
run() { env $2 timeout 300 python bench.py $SW $3 > $O/$1.json 2> $O/$1.err
  python - <<PY
import json
try:
    d=json.loads(open("$O/$1.json").read().strip().splitlines()[-1]); print("$1", d["value"], d["ms_per_step"], d["config"].get("launch"), d["config"]["loss"])
except Exception as e: print("$1 failed", e); print(open("$O/$1.err").read()[-1500:])
PY
}
run swin_small "A=1" ""
run swin_big "OFQ_NO_SMALL_TILES=1" ""
run swin_small2 "A=1" ""
