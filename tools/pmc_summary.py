#!/usr/bin/env python3
"""Per-kernel average of the PMC counters in a rocprofv3 rocpd database (run on the GPU box; prints a small table)."""
import sqlite3
import sys


def main(path):
    con = sqlite3.connect(path)
    cur = con.cursor()
    cols = [r[1] for r in cur.execute("pragma table_info(counters_collection)")]
    kcol = "kernel_name" if "kernel_name" in cols else [c for c in cols if "kernel" in c and "name" in c][0]
    ccol = "counter_name" if "counter_name" in cols else [c for c in cols if "counter" in c and "name" in c][0]
    vcol = "value" if "value" in cols else [c for c in cols if "value" in c][0]
    rows = cur.execute("select %s, %s, count(*), avg(%s), sum(%s) from counters_collection group by %s, %s order by 5 desc"
                       % (kcol, ccol, vcol, vcol, kcol, ccol)).fetchall()
    print("%-80s %-12s %7s %16s %16s" % ("kernel", "counter", "calls", "avg", "sum"))
    for r in rows[:40]:
        print("%-80s %-12s %7d %16.1f %16.1f" % (str(r[0])[:80], r[1], r[2], r[3], r[4]))


if __name__ == "__main__":
    main(sys.argv[1])
