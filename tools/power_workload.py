#!/usr/bin/env python3
"""Workload for tools/gpu/power_trace.sh: phases of ~SECS seconds each, with marks on time.monotonic() (CLOCK_MONOTONIC: the sampler process writes
the same clock): idle, the streaming dX GEMM (K = 2304, N = 384, 128 images) on 99 / 198 / 256 workgroups
in two and three planes, the whole default training step (graph replay).  Prints `phase name t_start_ms t_end_ms launches us_per_launch`."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from ofq_amd import ops, engine
from ofq_amd.quantization.utils import KDLossSoftandHard

SECS = float(os.environ.get("SECS", "4"))
T0 = 0.0                               # times are printed as time.monotonic() in ms: the sampler writes the same clock
M = 128 * 198


def phase(name, fn, sync=True):
    fn(); torch.cuda.synchronize()
    n, t0 = 0, time.monotonic()
    while time.monotonic() - t0 < SECS:
        for _ in range(20):
            fn()
        n += 20
        torch.cuda.synchronize()
    t1 = time.monotonic()
    print("phase %-34s %9.1f %9.1f %6d %8.1f" % (name, (t0 - T0) * 1e3, (t1 - T0) * 1e3, n, (t1 - t0) / n * 1e6), flush=True)
    time.sleep(1.0)


pr = torch.cuda.get_device_properties(0)
print("pci %04x:%02x:%02x.0" % (getattr(pr, "pci_domain_id", 0), getattr(pr, "pci_bus_id", 0), getattr(pr, "pci_device_id", 0)), flush=True)
torch.manual_seed(0)
dy = torch.randn(M, 2304, device="cuda") * 1e-3
qw = (2 * torch.randint(-2, 2, (2304, 384), device="cuda") + 1).to(torch.int8)
ks = torch.rand(2304, device="cuda") + 0.5
out = torch.empty(M, 384, device="cuda")
time.sleep(2.0)
t = time.monotonic()
print("phase %-34s %9.1f %9.1f %6d %8.1f" % ("idle", (t - 2.0 - T0) * 1e3, (t - T0) * 1e3, 0, 0.0), flush=True)
for planes, tr in ((2, ops.codes_transpose_f16), (3, ops.codes_transpose_bf16)):
    wT = tr(qw)
    for g in (99, 198, 256):
        phase("dX K=2304 planes=%d G=%d" % (planes, g), lambda: ops.qgemm_bf16s_nt_sk([(dy, wT, ks, 0.25)], out, wgs=g))
del dy, out
model = engine.build_student("deit_small_distilled_patch16_224", 2, 2, qk_reparam=True).cuda()
g = torch.Generator(device="cuda").manual_seed(42)
images = torch.randn(128, 3, 224, 224, device="cuda", generator=g)
target = torch.randint(0, 1000, (128,), device="cuda", generator=g)
soft = torch.randn(128, 1000, device="cuda", generator=g)
engine.setup_alpha(model, images)
model.train()
opt = engine.make_optimizer(model)
gs = engine.GraphedTrainStep(model, opt, KDLossSoftandHard(), warmup=2, alias_inputs=True)
for _ in range(4):
    gs(images, target, soft)
SECS = SECS * 2
phase("default training step (graph)", lambda: gs(images, target, soft))
