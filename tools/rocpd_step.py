#!/usr/bin/env python3
"""What ONE replayed training step launches, from a rocprofv3 rocpd (.db) kernel trace of bench.py.

The step is delimited by a kernel that runs exactly once per step (kd_loss_reduce_kernel, the last kernel of the loss); the last
`steps` complete periods of the trace are taken (the timed replays), and per period are printed: launches, time and idle time,
and the launches that are NOT this library's -- ATen element-wise / reduce / cat kernels, fills, runtime copies -- by name.
    python tools/rocpd_step.py trace.db [steps=10] [marker=kd_loss_reduce_kernel]"""
import sqlite3
import sys

OWN = ("absmax_", "adamw_", "assemble_tokens", "attn_f32", "cga_", "codes_transpose", "colsum_", "gelu_fwd", "gemm_bf16x3x3", "gemm_f32",
       "gemm_splitk", "input_pipeline", "kd_loss", "layernorm_", "lsq_", "nt_sk_", "permute_tokens", "qattn_", "qgemm_", "rowdot_",
       "softmax_lsq", "split_f32", "statsq_", "step_guard", "store_f32", "strided_sum")      # every __global__ of ofq_amd/csrc


def family(name):
    n = name[5:] if name.startswith("void ") else name
    if "rocclr_copyBuffer" in n:
        return "runtime copy (hipMemcpyAsync D2D)"
    if "rocclr_fillBuffer" in n:
        return "runtime fill (hipMemsetAsync)"
    if n.startswith("at::native") or n.startswith("(anonymous namespace)::softmax") or "at::native" in n[:40]:
        if "FillFunctor" in n:
            return "ATen fill"
        if "CatArray" in n:
            return "ATen cat"
        if "reduce_kernel" in n:
            return "ATen reduce"
        return "ATen element-wise"
    if n.startswith(OWN):
        return None
    return "other: " + n[:60]


def main(path, steps=10, marker="kd_loss_reduce_kernel"):
    cur = sqlite3.connect(path).cursor()
    cols = [r[1] for r in cur.execute("pragma table_info(kernels)").fetchall()]
    namecol = "name" if "name" in cols else [c for c in cols if "name" in c][0]
    rows = cur.execute("select start, end, %s from kernels order by start" % namecol).fetchall()
    marks = [i for i, r in enumerate(rows) if marker in r[2]]
    if len(marks) < steps + 1:
        raise SystemExit("only %d '%s' launches in the trace" % (len(marks), marker))
    marks = marks[-(steps + 1):]
    per = []
    for a, b in zip(marks[:-1], marks[1:]):
        seg = rows[a + 1:b + 1]
        busy = sum(e - s for s, e, _ in seg)
        reach, idle, big = rows[a][1], 0, 0
        for s, e, _ in seg:
            if s > reach:
                idle += s - reach
                big = max(big, s - reach)
            reach = max(reach, e)
        per.append((len(seg), (seg[-1][1] - rows[a][1]) / 1e6, busy / 1e6, idle / 1e3, big / 1e3, seg))
    print("%d periods between consecutive '%s' launches (the timed replays)" % (steps, marker))
    print("%8s %12s %18s %12s %16s" % ("launches", "period ms", "sum of kernels ms", "idle us", "largest gap us"))
    for n, span, busy, idle, big, _ in per:
        print("%8d %12.3f %18.3f %12.1f %16.1f" % (n, span, busy, idle, big))
    n = len(per)
    print("mean: %.1f launches, period %.3f ms, kernels %.3f ms, idle %.1f us (%.2f %% of the period), of which the largest gap (between two "
          "replays: the host's hipGraphLaunch) %.1f us" % (sum(p[0] for p in per) / n, sum(p[1] for p in per) / n, sum(p[2] for p in per) / n,
                                                          sum(p[3] for p in per) / n, 100.0 * sum(p[3] for p in per) / sum(p[1] for p in per) / 1e3,
                                                          sum(p[4] for p in per) / n))
    seg = per[-1][5]
    fam = {}
    for s, e, name in seg:
        f = family(name)
        if f is not None:
            d = fam.setdefault(f, {})
            k = (name[5:] if name.startswith("void ") else name)[:110]
            c = d.setdefault(k, [0, 0])
            c[0] += 1
            c[1] += e - s
    tot = sum(c[0] for d in fam.values() for c in d.values())
    print("launches of the last period that are not ofq_ kernels: %d of %d, %.1f us" % (tot, len(seg), sum(c[1] for d in fam.values() for c in d.values()) / 1e3))
    for f, d in sorted(fam.items()):
        print("  %-36s %4d launches %9.1f us" % (f, sum(c[0] for c in d.values()), sum(c[1] for c in d.values()) / 1e3))
        for k, c in sorted(d.items(), key=lambda kv: -kv[1][1]):
            print("      %4d x %8.1f us  %s" % (c[0], c[1] / 1e3 / c[0], k))


if __name__ == "__main__":
    main(sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 10, sys.argv[3] if len(sys.argv) > 3 else "kd_loss_reduce_kernel")
