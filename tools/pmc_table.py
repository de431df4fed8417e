#!/usr/bin/env python3
"""Per-kernel table of ALL counters of one or more rocprofv3 --pmc passes (rocpd databases): sum over the launches of a kernel
divided by the number of launches.  Usage: pmc_table.py <db> [<db> ...] [--match substring]"""
import collections
import sqlite3
import sys


def load(path, per, calls):
    cur = sqlite3.connect(path).cursor()
    cols = [r[1] for r in cur.execute("pragma table_info(counters_collection)")]
    kcol = "kernel_name" if "kernel_name" in cols else [c for c in cols if "kernel" in c and "name" in c][0]
    ccol = "counter_name" if "counter_name" in cols else [c for c in cols if "counter" in c and "name" in c][0]
    vcol = "value" if "value" in cols else [c for c in cols if "value" in c][0]
    dcol = [c for c in cols if "dispatch" in c and "id" in c][0]
    for k, c, n, sm in cur.execute("select %s, %s, count(distinct %s), sum(%s) from counters_collection group by %s, %s"
                                   % (kcol, ccol, dcol, vcol, kcol, ccol)):
        per[str(k)][c] = float(sm) / max(n, 1)
        calls[str(k)] = n


def main(argv):
    match = None
    if "--match" in argv:
        i = argv.index("--match")
        match = argv[i + 1]
        argv = argv[:i] + argv[i + 2:]
    per, calls = collections.defaultdict(dict), {}
    for p in argv:
        load(p, per, calls)
    names = sorted({c for d in per.values() for c in d})
    for k in sorted(per, key=lambda k: -per[k].get("GRBM_GUI_ACTIVE", 0.0) * calls[k]):
        if match and match not in k:
            continue
        print("%s   (%d launches; per launch)" % (k[:150], calls[k]))
        for c in names:
            if c in per[k]:
                print("    %-32s %18.1f" % (c, per[k][c]))


if __name__ == "__main__":
    main(sys.argv[1:])
