#!/usr/bin/env python3
"""power.csv (tools/power_sampler.py) x phases.txt (tools/power_workload.py) -> per phase: mean / max socket power and mean / min
shader clock of the card whose power moved most (both files carry time.monotonic() in ms: one clock for every process of the box;
the first 300 ms and the last 100 ms of a phase are left out)."""
import sys
csv, phases = sys.argv[1], sys.argv[2]
rows = [l.strip() for l in open(csv) if l.strip()]
head = [l for l in rows if l.startswith("#")]
data = [[float(v) for v in l.split(",")] for l in rows if l[0].isdigit()]
ncard = (len(data[0]) - 1) // 2
swing = [max(r[1 + 2 * i] for r in data) - min(r[1 + 2 * i] for r in data) for i in range(ncard)]
c = max(range(ncard), key=lambda i: swing[i])
# the node is shared: other tenants' GPUs show up in sysfs (and swing as well) -- take the card the workload names by PCI address
pcis = [l.split(":", 1)[1].split() for l in head if l.startswith("# pci:")]
mine = [l.split()[1] for l in open(phases) if l.startswith("pci ")]
if pcis and mine and mine[0] in pcis[0]:
    c = pcis[0].index(mine[0])
print("\n".join(head))
print("card index %d (power swing %.0f W; the other cards: %s)" % (c, swing[c], " ".join("%.0f" % s for i, s in enumerate(swing) if i != c)))
print("%-66s %8s %8s %8s %8s %8s %10s" % ("phase", "P_mean_W", "P_max_W", "f_mean", "f_min", "samples", "us/launch"))
for l in open(phases):
    if not l.startswith("phase"):
        continue
    parts = l.split()
    us, n, t1, ts = float(parts[-1]), int(parts[-2]), float(parts[-3]), float(parts[-4])
    name = " ".join(parts[1:-4])
    sel = [r for r in data if ts + 300 <= r[0] <= t1 - 100]
    if not sel:
        print("%-66s (no samples)" % name)
        continue
    P = [r[1 + 2 * c] for r in sel]
    F = [r[2 + 2 * c] for r in sel]
    print("%-66s %8.0f %8.0f %8.0f %8.0f %8d %10.1f" % (name, sum(P) / len(P), max(P), sum(F) / len(F), min(F), len(sel), us))
