#!/usr/bin/env python3
"""Counters of every dispatch in a rocprofv3 --pmc rocpd database, in launch order (for probe binaries that launch known shapes)."""
import sqlite3, sys, collections
con = sqlite3.connect(sys.argv[1]); cur = con.cursor()
cols = [r[1] for r in cur.execute("pragma table_info(counters_collection)")]
print(cols)
kcol = "kernel_name" if "kernel_name" in cols else [c for c in cols if "kernel" in c and "name" in c][0]
ccol = "counter_name" if "counter_name" in cols else [c for c in cols if "counter" in c and "name" in c][0]
vcol = "value" if "value" in cols else [c for c in cols if "value" in c][0]
dcol = [c for c in cols if "dispatch" in c and "id" in c][0]
gcol = [c for c in cols if "grid" in c][0] if any("grid" in c for c in cols) else None
rows = cur.execute("select %s, %s, %s, %s, sum(%s) from counters_collection group by %s, %s order by %s"
                   % (dcol, kcol, gcol or "0", ccol, vcol, dcol, ccol, dcol)).fetchall()
d = collections.OrderedDict()
for disp, k, g, c, v in rows:
    d.setdefault((disp, k, g), {})[c] = v
names = sorted({c for v in d.values() for c in v})
print("%-6s %-44s %-10s " % ("disp", "kernel", "grid") + " ".join("%16s" % n[-16:] for n in names))
for (disp, k, g), v in d.items():
    if "copyBuffer" in k or "fillBuffer" in k: continue
    print("%-6s %-44s %-10s " % (disp, str(k)[:44], g) + " ".join("%16.0f" % v.get(n, 0) for n in names))
