#!/usr/bin/env python3
"""Workload for tools/gpu/power_mfma.sh: what matrix-core rate does the part SUSTAIN under its power cap when a kernel issues
nothing but MFMAs (tools/probe/mfma_power_probe.hip)?  Phases of ~SECS seconds: fp16 32x32x16 and int8 32x32x32 loops on 128 / 198 /
256 CUs at one and two waves per SIMD, then the library fp16 GEMM and the two-plane dX kernel for comparison.  Prints
`phase name t_start_ms t_end_ms launches us_per_launch` (clock: time.monotonic, shared with tools/power_sampler.py) and the
achieved rate per phase in the name."""
import ctypes, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from ofq_amd import ops

SECS = float(os.environ.get("SECS", "3"))
lib = ctypes.CDLL(os.path.join(ROOT, "tools", "probe", "bin", "libmfma_power.so"))
lib.mfma_power_launch.argtypes = [ctypes.c_int, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
out = torch.empty(2048 * 512, device="cuda")
ITERS = 2000                                     # x 24 MFMAs per wave and launch


def phase(name, fn, ops_per_launch):
    fn(); torch.cuda.synchronize()
    n, t0 = 0, time.monotonic()
    while time.monotonic() - t0 < SECS:
        for _ in range(5):
            fn()
        n += 5
        torch.cuda.synchronize()
    t1 = time.monotonic()
    us = (t1 - t0) / n * 1e6
    print("phase %-58s %9.1f %9.1f %6d %8.1f" % ("%s [%.2f Pop/s]" % (name, ops_per_launch / us / 1e9), t0 * 1e3, t1 * 1e3, n, us), flush=True)
    time.sleep(1.0)


pr = torch.cuda.get_device_properties(0)
print("pci %04x:%02x:%02x.0" % (getattr(pr, "pci_domain_id", 0), getattr(pr, "pci_bus_id", 0), getattr(pr, "pci_device_id", 0)), flush=True)
time.sleep(2.0)
t = time.monotonic()
print("phase %-58s %9.1f %9.1f %6d %8.1f" % ("idle", (t - 2.0) * 1e3, t * 1e3, 0, 0.0), flush=True)
st = torch.cuda.current_stream().cuda_stream
for kind, nm, opw in ((0, "fp16 32x32x16", 32768.0), (1, "int8 32x32x32", 65536.0)):
    for blocks in (128, 198, 256):
        for threads in (256, 512):
            waves = blocks * threads // 64
            phase("%s only, %d CUs x %d wave(s)/SIMD" % (nm, blocks, threads // 256),
                  lambda: lib.mfma_power_launch(kind, out.data_ptr(), blocks, threads, ITERS, st), waves * 24.0 * ITERS * opw)
M = 128 * 198
A = torch.randn(M, 4608, device="cuda", dtype=torch.float16)
B = torch.randn(384, 4608, device="cuda", dtype=torch.float16)
phase("library fp16 GEMM 25344 x 384 x 4608", lambda: torch.matmul(A, B.t()), 2.0 * M * 384 * 4608)
A2 = torch.randn(8192, 8192, device="cuda", dtype=torch.float16)
phase("library fp16 GEMM 8192^3", lambda: torch.matmul(A2, A2), 2.0 * 8192 ** 3)
dy = torch.randn(M, 2304, device="cuda") * 1e-3
wT = ops.codes_transpose_f16((2 * torch.randint(-2, 2, (2304, 384), device="cuda") + 1).to(torch.int8))
ks = torch.rand(2304, device="cuda") + 0.5
o = torch.empty(M, 384, device="cuda")
phase("two-plane dX kernel K=2304 (matrix-core work)", lambda: ops.qgemm_bf16s_nt_sk([(dy, wT, ks, 0.25)], o, wgs=198), 2.0 * M * 384 * 4608)
