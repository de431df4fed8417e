#!/usr/bin/env python3
"""Summarise a rocprofv3 rocpd (.db) kernel trace: per-kernel calls / total / avg / % (like --stats CSV)."""
import sqlite3
import sys


def main(path, top=40, by_grid=False):
    con = sqlite3.connect(path)
    cur = con.cursor()
    cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
    name_col = "name" if "name" in cols else [c for c in cols if "name" in c][0]
    if by_grid:     # one line per (kernel, grid) so the shapes of one kernel can be told apart
        gcol = [c for c in cols if "grid" in c and c.endswith("x")]
        if not gcol:
            print("no grid column among", cols)
            return
        name_col = "substr(%s, 1, 60) || ' grid=' || %s" % (name_col, gcol[0])
    rows = cur.execute("select %s, count(*), sum(end - start), avg(end - start), min(end-start), max(end-start) "
                       "from kernels group by %s order by 3 desc" % (name_col, name_col)).fetchall()
    total = sum(r[2] for r in rows)
    print("%-90s %7s %12s %10s %10s %10s %6s" % ("kernel", "calls", "total_us", "avg_us", "min_us", "max_us", "%"))
    for r in rows[:top]:
        print("%-90s %7d %12.1f %10.1f %10.1f %10.1f %6.2f" % (r[0][:90], r[1], r[2] / 1e3, r[3] / 1e3, r[4] / 1e3,
                                                              r[5] / 1e3, 100.0 * r[2] / total))
    print("TOTAL kernel time %.1f ms over %d kernels" % (total / 1e6, sum(r[1] for r in rows)))


if __name__ == "__main__":
    main(sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 40, by_grid=len(sys.argv) > 3 and sys.argv[3] == "grid")
