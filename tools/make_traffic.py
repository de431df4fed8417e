#!/usr/bin/env python3
"""HBM traffic per kernel class from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; separate passes, rocpd .db):

    python tools/make_traffic.py fetch.db write.db steps out.json [fetch2.db write2.db extra_steps]

Counters are KB per dispatch; FETCH_SIZE is doubled (gfx950 tallies 128-B read requests at 64 B: MI355X_MICROARCH.md,
HBM section).  Classes are the ones bench.py times (ops._Timed).  The file carries the content hash of the kernel sources
(ofq_amd/build.py:source_hash): bench.py refuses traffic measured on other kernels."""
import json
import os
import sqlite3
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

# bench.py's timer classes (first word of the ops._Timed name) -> the kernel(s) behind them
CLASSES = [("qgemm_bf16s_nt_wide", ("qgemm_bf16s_nt_wide_kernel", "qgemm_bf16s_nt_wide_sk_kernel")),     # classic + streaming / two-segment
           ("qgemm_bf16s_tn_wide_group", ("qgemm_bf16s_tn_wide_group_kernel",)),
           ("qgemm_bf16s_tn_wide_stream", ("qgemm_bf16s_tn_wide_stream_kernel",)),
           ("qgemm_bf16s_tn_wide", ("qgemm_bf16s_tn_wide_kernel",)),
           ("qgemm_bf16s_nn_wide", ("qgemm_bf16s_nn_wide_kernel",)),
           ("qgemm_bf16s_tn", ("qgemm_bf16s_tn_kernel",)),
           ("qgemm_bf16s_nt", ("qgemm_bf16s_nt_kernel",)),
           ("qgemm_bf16s_nn", ("qgemm_bf16s_nn_kernel",)),
           ("qgemm_i8_nt", ("qgemm_i8_nt_kernel",)),
           ("qgemm_i8_lsqbwd", ("qgemm_i8_lsqbwd_kernel",)),
           ("qattn_scores_softmax", ("qattn_scores_softmax_kernel",)),
           ("qattn_dp_softmax_bwd", ("qattn_dp_softmax_bwd_kernel",)),
           ("gemm_f32", ("gemm_f32_fast_kernel", "gemm_f32_kernel"))]


def mfma_util(path):
    """kernel name -> MfmaUtil % of a `--pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE ...` pass (tools/pmc_mfma.py's formula
    and calibration: raw = 100 * busy / (GRBM_GUI_ACTIVE * 1024 SIMDs), the back-to-back MFMA probe reads raw = 11.6)."""
    cur = sqlite3.connect(path).cursor()
    cols = [r[1] for r in cur.execute("pragma table_info(counters_collection)")]
    kcol = "kernel_name" if "kernel_name" in cols else [c for c in cols if "kernel" in c and "name" in c][0]
    ccol = "counter_name" if "counter_name" in cols else [c for c in cols if "counter" in c and "name" in c][0]
    vcol = "value" if "value" in cols else [c for c in cols if "value" in c][0]
    d = {}
    for k, c, sm in cur.execute("select %s, %s, sum(%s) from counters_collection group by %s, %s" % (kcol, ccol, vcol, kcol, ccol)):
        d.setdefault(str(k), {})[c] = float(sm)
    return d


def per_kernel(path, counter):
    cur = sqlite3.connect(path).cursor()
    cols = [r[1] for r in cur.execute("pragma table_info(counters_collection)")]
    kcol = "kernel_name" if "kernel_name" in cols else [c for c in cols if "kernel" in c and "name" in c][0]
    ccol = "counter_name" if "counter_name" in cols else [c for c in cols if "counter" in c and "name" in c][0]
    vcol = "value" if "value" in cols else [c for c in cols if "value" in c][0]
    dcol = [c for c in cols if "dispatch" in c and "id" in c][0]
    out = {}
    q = ("select %s, count(distinct %s), sum(%s) from counters_collection where %s = ? group by %s"
         % (kcol, dcol, vcol, ccol, kcol))
    for k, n, sm in cur.execute(q, (counter,)):
        out[str(k)] = (int(n), float(sm))
    return out


def main(fetch_db, write_db, steps, out_path, fetch_db2=None, write_db2=None, extra_steps=None, mfma_db=None):
    """fetch_db2 / write_db2: the same passes with `extra_steps` more timed steps -- their difference is the traffic of
    exactly that many steady-state steps (setup_alpha's initialisation kernels and the warm-up cancel)."""
    from ofq_amd import build
    steps = float(steps)
    f, w = per_kernel(fetch_db, "FETCH_SIZE"), per_kernel(write_db, "WRITE_SIZE")
    res = {"_source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, --kernel-trace) over `python3 bench.py "
                      "--no-graph` (counter collection serialises the launches; same kernels as the replayed graph); KB per "
                      "dispatch, FETCH_SIZE doubled (gfx950 counts 128-B requests as 64 B); traffic = 2*FETCH + WRITE",
           "source_hash": build.source_hash(), "steps_in_passes": steps}
    tot_f = sum(v[1] for v in f.values())
    tot_w = sum(v[1] for v in w.values())
    res["whole_step"] = {"read_GB": round(2 * tot_f * 1024 / steps / 1e9, 2), "written_GB": round(tot_w * 1024 / steps / 1e9, 2),
                         "total_GB": round((2 * tot_f + tot_w) * 1024 / steps / 1e9, 2),
                         "note": "all kernels of the profiled process divided by the number of steps (warm-up, setup_alpha and "
                                 "the timed steps together: an upper bound per step)"}
    if fetch_db2 and write_db2:
        n = float(extra_steps)
        f2, w2 = per_kernel(fetch_db2, "FETCH_SIZE"), per_kernel(write_db2, "WRITE_SIZE")
        df = sum(v[1] for v in f2.values()) - tot_f
        dw = sum(v[1] for v in w2.values()) - tot_w
        res["steady_step"] = {"read_GB": round(2 * df * 1024 / n / 1e9, 2), "written_GB": round(dw * 1024 / n / 1e9, 2),
                              "total_GB": round((2 * df + dw) * 1024 / n / 1e9, 2),
                              "note": "difference of two pass pairs that differ by %d timed steps, per step" % int(n)}
        keys = set(f) | set(w) | set(f2) | set(w2)
        g = lambda d, k: d.get(k, (0, 0.0))[1]
        topd = sorted(((2 * (g(f2, k) - g(f, k)) + (g(w2, k) - g(w, k))) * 1024 / n / 1e9, k) for k in keys)[::-1][:25]
        res["top_kernels_GB_per_steady_step"] = [[k[:70], round(v, 2)] for v, k in topd]
    for cls, prefixes in CLASSES:
        nf = sum(v[0] for k, v in f.items() if any(p in k for p in prefixes))
        sf = sum(v[1] for k, v in f.items() if any(p in k for p in prefixes))
        nw = sum(v[0] for k, v in w.items() if any(p in k for p in prefixes))
        sw = sum(v[1] for k, v in w.items() if any(p in k for p in prefixes))
        if nf and nw:
            res[cls] = {"launches": nf, "fetch_size_kb_raw_per_launch": round(sf / nf, 1), "write_size_kb_per_launch": round(sw / nw, 1),
                        "traffic_bytes_per_launch": int((2 * sf / nf + sw / nw) * 1024)}
            if mfma_db:
                mu = mfma_util(mfma_db)
                busy = sum(v.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) for k, v in mu.items() if any(p in k for p in prefixes))
                gui = sum(v.get("GRBM_GUI_ACTIVE", 0.0) for k, v in mu.items() if any(p in k for p in prefixes))
                if gui > 0:
                    res[cls]["mfma_util_pmc"] = round(100.0 * busy / (gui * 1024) / 0.116, 1)
    top = sorted(((2 * f.get(k, (0, 0))[1] + w.get(k, (0, 0))[1]) * 1024 / steps / 1e9, k) for k in set(f) | set(w))[::-1][:25]
    res["top_kernels_GB_per_step"] = [[k[:70], round(g, 2)] for g, k in top]
    with open(out_path, "w") as fh:
        json.dump(res, fh, indent=1)
    print(json.dumps(res["whole_step"]))
    if "steady_step" in res:
        print("steady:", json.dumps(res["steady_step"]))
    for k, g in res["top_kernels_GB_per_step"][:12]:
        print("%6.2f GB/step  %s" % (g, k))


if __name__ == "__main__":
    main(*sys.argv[1:9])
