#!/bin/bash
set -u
O=gpurun_out/r02o; mkdir -p $O
export TMPDIR=/tmp
cd /tmp
timeout 400 rocprofv3 --kernel-trace -d $GRAFT_REPO_ROOT/$O/prof -o p -- python3 $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-roofline-events > $GRAFT_REPO_ROOT/$O/prof.log 2>&1; echo "rocprof rc=$?"
cd $GRAFT_REPO_ROOT
db=$(find $O/prof -name "*.db" | head -1)
python - $db <<'PY'
import sqlite3, sys
cur = sqlite3.connect(sys.argv[1]).cursor()
cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
print([c for c in cols if "grid" in c or "workgroup" in c or "lds" in c.lower()])
q = "select name, grid_x, grid_y, count(*), avg(end-start)/1e3, min(end-start)/1e3 from kernels where name like '%wide%' or name like '%i8_nt_kernel<0%' group by name, grid_x, grid_y order by 4*5 desc"
for r in cur.execute(q):
    print("%-58s gx=%-8d gy=%-6d calls=%-5d avg=%8.1f us min=%8.1f" % (r[0][:58], r[1], r[2], r[3], r[4], r[5]))
PY
find $O/prof -name "*.db" -delete
