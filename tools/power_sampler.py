#!/usr/bin/env python3
"""A second instrument for the "the wide GEMMs are power-bound" claim: socket power and shader clock from the amdgpu
hwmon files (power1_input, freq1_input), every PERIOD ms, WITHOUT touching the GPU runtime -- started as a sibling process before
the workload (tools/gpu/power_trace.sh) and stopped by it.  Every hwmon directory of the box is sampled (the sysfs tree shows
all GPUs of the node); the one whose power moves most is the GPU the workload ran on.

    python tools/power_sampler.py out.csv [period_ms]          # runs until SIGTERM / SIGINT; then writes out.csv
Columns: t_ms (time.monotonic, shared with the workload process), then per card: power_W, sclk_MHz."""
import glob, os, signal, sys, time

out = sys.argv[1]
period = float(sys.argv[2]) / 1e3 if len(sys.argv) > 2 else 0.010
dirs = sorted(glob.glob("/sys/class/drm/card*/device/hwmon/hwmon*"))
dirs = [d for d in dirs if os.path.exists(os.path.join(d, "power1_input")) and os.path.exists(os.path.join(d, "freq1_input"))]
files = [(open(os.path.join(d, "power1_input")), open(os.path.join(d, "freq1_input"))) for d in dirs]
caps = []
for d in dirs:
    try:
        caps.append(int(open(os.path.join(d, "power1_cap")).read()) / 1e6)
    except OSError:
        caps.append(float("nan"))
rows, stop = [], [False]
signal.signal(signal.SIGTERM, lambda *a: stop.__setitem__(0, True))
signal.signal(signal.SIGINT, lambda *a: stop.__setitem__(0, True))
t0 = time.monotonic()
while not stop[0]:
    t = time.monotonic()
    r = [t * 1e3]                       # CLOCK_MONOTONIC in ms: the same clock in every process of the box
    for fp, ff in files:
        try:
            fp.seek(0); ff.seek(0)
            r += [int(fp.read()) / 1e6, int(ff.read()) / 1e6]
        except (OSError, ValueError):
            r += [float("nan"), float("nan")]
    rows.append(r)
    dt = period - (time.monotonic() - t)
    if dt > 0:
        time.sleep(dt)
with open(out, "w") as fh:
    pci = [os.path.basename(os.path.realpath(os.path.join(d, "device"))) for d in dirs]        # e.g. 0000:75:00.0
    fh.write("# cards: %s\n# pci: %s\n# power caps (W): %s\n" % (" ".join(d.split("/")[4] for d in dirs), " ".join(pci),
                                                                " ".join("%.0f" % c for c in caps)))
    fh.write("t_ms," + ",".join("p%d_W,f%d_MHz" % (i, i) for i in range(len(dirs))) + "\n")
    for r in rows:
        fh.write(",".join("%.1f" % v for v in r) + "\n")
print("power_sampler: %d samples over %.1f s from %d hwmon directories" % (len(rows), (time.monotonic() - t0), len(dirs)))
