#!/bin/bash
set -u
O=gpurun_out/r02g; mkdir -p $O
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q > $O/gpu_tests.txt 2>&1; echo "gpu tests rc=$?"
tail -40 $O/gpu_tests.txt | cut -c1-300
