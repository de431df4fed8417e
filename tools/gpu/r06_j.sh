#!/bin/bash
set -u
O=gpurun_out/r06_j; mkdir -p $O
cd $GRAFT_REPO_ROOT
python tools/op_trace.py > $O/op_trace.txt 2>&1; tail -32 $O/op_trace.txt | cut -c1-330
