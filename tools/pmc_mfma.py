#!/usr/bin/env python3
"""Per-kernel MFMA utilisation from a rocprofv3 --pmc pass (rocpd database):
   raw = 100 * SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE * 1024 SIMDs) as in the gfx94x derived metric; calibration:
   tools/probe/mfma_rate_probe (back-to-back v_mfma_f32_32x32x16_bf16 on every SIMD = 100 % by construction) reads raw = 11.6
   under the same pass on this pool, so MfmaUtil = raw / 0.116.  Also the share of wave time spent issuing / stalled at
   issue / parked (SQ_ACTIVE_INST_ANY, SQ_WAIT_INST_ANY, SQ_WAIT_ANY over SQ_WAVE_CYCLES; a SIMD hosts 2-5 waves)."""
import sqlite3, sys, collections
con = sqlite3.connect(sys.argv[1])
cur = con.cursor()
cols = [r[1] for r in cur.execute("pragma table_info(counters_collection)")]
kcol = "kernel_name" if "kernel_name" in cols else [c for c in cols if "kernel" in c and "name" in c][0]
ccol = "counter_name" if "counter_name" in cols else [c for c in cols if "counter" in c and "name" in c][0]
vcol = "value" if "value" in cols else [c for c in cols if "value" in c][0]
d = collections.defaultdict(dict)
n = {}
for k, c, cnt, sm in cur.execute("select %s, %s, count(*), sum(%s) from counters_collection group by %s, %s" % (kcol, ccol, vcol, kcol, ccol)):
    d[k][c] = sm
    n[k] = cnt
SIMDS = 256 * 4
rows = []
for k, v in d.items():
    if "GRBM_GUI_ACTIVE" not in v or not v.get("GRBM_GUI_ACTIVE"):
        continue
    gui = v["GRBM_GUI_ACTIVE"]
    wc = v.get("SQ_WAVE_CYCLES", 0) or 1
    rows.append((gui, k, n[k], 100.0 * v.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / (gui * SIMDS),
                 100.0 * v.get("SQ_ACTIVE_INST_ANY", 0) / wc, 100.0 * v.get("SQ_WAIT_INST_ANY", 0) / wc,
                 100.0 * v.get("SQ_WAIT_ANY", 0) / wc, 100.0 * v.get("SQ_ACTIVE_INST_VALU", 0) / wc))
rows.sort(reverse=True)
print(__doc__)
print("%-86s %6s %6s %9s %8s %8s %8s %8s" % ("kernel", "calls", "raw", "MfmaUtil%", "issue%", "istall%", "parked%", "valu%"))
for r in rows[:28]:
    print("%-86s %6d %6.1f %9.1f %8.1f %8.1f %8.1f %8.1f" % (str(r[1])[:86], r[2], r[3], r[3] / 0.116, r[4], r[5], r[6], r[7]))
