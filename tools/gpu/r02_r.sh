#!/bin/bash
set -u
O=gpurun_out/r02r; mkdir -p $O
export TMPDIR=/tmp
cd /tmp
timeout 400 rocprofv3 --kernel-trace -d $GRAFT_REPO_ROOT/$O/prof -o p -- python3 $GRAFT_REPO_ROOT/tools/teacher_profile.py > $GRAFT_REPO_ROOT/$O/prof.log 2>&1; echo "rocprof rc=$?"
cd $GRAFT_REPO_ROOT
grep "forward" $O/prof.log
db=$(find $O/prof -name "*.db" | head -1)
python - $db <<'PY'
import sqlite3, sys
cur = sqlite3.connect(sys.argv[1]).cursor()
q = "select name, grid_x, grid_y, grid_z, count(*), avg(end-start)/1e3, sum(end-start)/1e3 from kernels group by name, grid_x, grid_y, grid_z order by 7 desc limit 24"
for r in cur.execute(q):
    print("%-64s g=%d,%d,%d calls=%-4d avg=%8.1f us total=%9.1f" % (r[0][:64], r[1], r[2], r[3], r[4], r[5], r[6]))
PY
find $O/prof -name "*.db" -delete
