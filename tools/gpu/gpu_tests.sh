#!/bin/bash
set -u
O=gpurun_out/r03_tests; mkdir -p $O
timeout 2400 python -m pytest tests -m gpu -q -x > $O/tests.txt 2>&1; echo "tests rc=$?"; tail -15 $O/tests.txt
