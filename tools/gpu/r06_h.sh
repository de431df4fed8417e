#!/bin/bash
set -u
O=gpurun_out/r06_h; mkdir -p $O
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
timeout 3000 python -m pytest tests -q -m gpu > $O/gpu_tests.txt 2>&1; echo "gpu tests rc=$?"; tail -25 $O/gpu_tests.txt | cut -c1-300
