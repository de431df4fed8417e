#!/bin/bash
# Which property of the attention prep's row-dot makes it fail on a shared GPU?  Variant libraries built with -DRD16_VARIANT=v
# (csrc/qgemm_codes.hip): 3 = round 5's form (control), 2 = round 5's form + row pointers kept live, 1 = round 6's form without the
# keep-alive, default library = round 6's form as shipped.  The RD16_VARIANT switch exists in commit a29d053 only (HEAD keeps the
# shipped form): build the variant libraries from that commit's csrc into tools/probe/bin/ before running this.
set -u
O=gpurun_out/r06_variants; mkdir -p $O
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
for v in 3 2 1 0; do
  if [ $v = 0 ]; then unset OFQ_HIP_LIB; else export OFQ_HIP_LIB=$GRAFT_REPO_ROOT/tools/probe/bin/libofq_rd16_v$v.so; fi
  ( MODE=solo REPS=700 CFGS="nodp" timeout 600 python tools/two_rank_trace.py ) > $O/v$v.txt 2>&1; echo "variant $v rc=$?"
  grep -E "cfg" $O/v$v.txt | cut -c1-200
done
unset OFQ_HIP_LIB
( MODE=ranks REPS=200 CFGS="base" timeout 600 python tools/two_rank_trace.py ) > $O/ranks_base.txt 2>&1; echo "ranks rc=$?"
grep -E "cfg|rep " $O/ranks_base.txt | cut -c1-300 | head
