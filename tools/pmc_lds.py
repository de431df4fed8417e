#!/usr/bin/env python3
"""LDS bank-conflict share per kernel from a rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS
GRBM_GUI_ACTIVE pass: conflict cycles / active LDS cycles, LDS-active share of the kernel's time (per CU)."""
import sqlite3
import sys


def main(path, top=14):
    cur = sqlite3.connect(path).cursor()
    cols = [r[1] for r in cur.execute("pragma table_info(counters_collection)")]
    kcol = "kernel_name" if "kernel_name" in cols else [c for c in cols if "kernel" in c and "name" in c][0]
    ccol = "counter_name" if "counter_name" in cols else [c for c in cols if "counter" in c and "name" in c][0]
    vcol = "value" if "value" in cols else [c for c in cols if "value" in c][0]
    dcol = [c for c in cols if "dispatch" in c and "id" in c][0]
    per = {}
    for k, c, n, sm in cur.execute("select %s, %s, count(distinct %s), sum(%s) from counters_collection group by %s, %s"
                                   % (kcol, ccol, dcol, vcol, kcol, ccol)):
        per.setdefault(str(k), {})[c] = (n, float(sm))
    rows = []
    for k, d in per.items():
        if "SQ_LDS_IDX_ACTIVE" not in d or "GRBM_GUI_ACTIVE" not in d:
            continue
        act, conf = d["SQ_LDS_IDX_ACTIVE"][1], d.get("SQ_LDS_BANK_CONFLICT", (0, 0.0))[1]
        gui = d["GRBM_GUI_ACTIVE"][1]
        rows.append((gui, k, d["GRBM_GUI_ACTIVE"][0], 100.0 * conf / act if act else 0.0, 100.0 * act / (gui * 256) if gui else 0.0,
                     d.get("SQ_INSTS_LDS", (0, 0.0))[1] / max(d["GRBM_GUI_ACTIVE"][0], 1)))
    rows.sort(reverse=True)
    print("%-84s %6s %10s %12s %14s" % ("kernel", "calls", "conflict%", "LDS busy%", "LDS inst/call"))
    for gui, k, n, cp, lp, ni in rows[:top]:
        print("%-84s %6d %10.1f %12.1f %14.0f" % (k[:84], n, cp, lp, ni))


if __name__ == "__main__":
    main(sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 14)
