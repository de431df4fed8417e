#!/bin/bash
set -u
O=gpurun_out/r03_d; mkdir -p $O
timeout 900 python -m pytest tests/test_prod_gpu.py -q -x -s > $O/prod.txt 2>&1; echo "prod rc=$?"; grep "golden:\|passed\|failed" $O/prod.txt
timeout 1500 python -m pytest tests/test_depth12_gpu.py -q -s > $O/d12.txt 2>&1; echo "d12 rc=$?"; grep -v "^$" $O/d12.txt | tail -90
