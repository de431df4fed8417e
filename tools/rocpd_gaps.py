#!/usr/bin/env python3
"""Idle time between kernels in a rocprofv3 rocpd (.db) kernel trace: over the last `frac` of the dispatches (the timed
steps of bench.py) print span, busy time, idle time and the distribution of the gaps between consecutive kernels.
    python tools/rocpd_gaps.py trace.db [frac=0.5]"""
import sqlite3
import sys


def main(path, frac=0.5):
    cur = sqlite3.connect(path).cursor()
    rows = cur.execute("select start, end from kernels order by start").fetchall()
    n = len(rows)
    rows = rows[int(n * (1.0 - frac)):]
    busy = sum(e - s for s, e in rows)
    span = max(e for _, e in rows) - rows[0][0]
    gaps, reach = [], rows[0][1]
    for s, e in rows[1:]:
        if s > reach:
            gaps.append(s - reach)
        reach = max(reach, e)
    idle = sum(gaps)
    gaps.sort()
    q = lambda p: gaps[min(len(gaps) - 1, int(p * len(gaps)))] / 1e3 if gaps else 0.0
    print("kernels %d  span %.2f ms  busy(sum of durations) %.2f ms  idle %.2f ms (%.1f%% of span)"
          % (len(rows), span / 1e6, busy / 1e6, idle / 1e6, 100.0 * idle / span))
    print("gaps: n %d  mean %.2f us  p50 %.2f  p90 %.2f  p99 %.2f  max %.2f us"
          % (len(gaps), idle / max(len(gaps), 1) / 1e3, q(0.5), q(0.9), q(0.99), gaps[-1] / 1e3 if gaps else 0.0))
    big = [g for g in gaps if g > 20000]
    print("gaps > 20 us: %d, together %.2f ms" % (len(big), sum(big) / 1e6))


if __name__ == "__main__":
    main(sys.argv[1], float(sys.argv[2]) if len(sys.argv) > 2 else 0.5)
