#!/bin/bash
set -u
O=gpurun_out/r03_c; mkdir -p $O
timeout 120 tools/probe/bin/concurrency_probe > $O/conc.txt 2>&1; cat $O/conc.txt
timeout 900 python -m pytest tests/test_prod_gpu.py -q -x -s > $O/prod.txt 2>&1; echo "prod rc=$?"; grep -v "^$" $O/prod.txt | tail -40
timeout 900 python -m pytest tests/test_modules_gpu.py -q -x > $O/mods.txt 2>&1; echo "mods rc=$?"; tail -5 $O/mods.txt
