#!/usr/bin/env python3
"""Which C-ABI entries one training step calls, how often and with which integer arguments (the shapes a kernel family sees).

Wraps every function of the loaded libofq_hip.so (ofq_amd/_lib.SIGNATURES) for ONE eager step and counts the calls per
(entry, tuple of its integer / float scalar arguments); pointers are printed as '*' ('0' when NULL).
    MODEL=swin_t BITS=3 QKR=1 BATCH=128 ONLY=ofq_qgemm_bf16s_tn,ofq_qattn_dv_bf16s python tools/lib_call_shapes.py"""
import collections
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from ofq_amd import _lib, engine, ops
from ofq_amd.quantization.utils import KDLossSoftandHard

name = os.environ.get("MODEL", "deit_small_distilled_patch16_224")
bits, qkr, B = int(os.environ.get("BITS", "2")), os.environ.get("QKR", "1") == "1", int(os.environ.get("BATCH", "128"))
only = [s for s in os.environ.get("ONLY", "").split(",") if s]
torch.manual_seed(0)
model = engine.build_student(name, bits, bits, qk_reparam=qkr).cuda()
batch = (torch.randn(B, 3, 224, 224, device="cuda"), torch.randint(0, 1000, (B,), device="cuda"), torch.randn(B, 1000, device="cuda"))
engine.setup_alpha(model, batch[0][:16])
model.train()
opt = engine.make_optimizer(model)
crit = KDLossSoftandHard()
for _ in range(2):
    engine.train_step(model, opt, *batch, crit)
torch.cuda.synchronize()

L = ops.lib()
counts = collections.Counter()
saved = {}
for fn, (res, args) in _lib.SIGNATURES.items():
    if only and fn not in only:
        continue
    f = getattr(L, fn)
    saved[fn] = f
    scal = [i for i, t in enumerate(args) if t in (_lib.i64, _lib.i32, _lib.f32, _lib.f64, _lib.sz)]
    ptrs = [i for i, t in enumerate(args) if t is _lib.vp]

    def wrapped(*a, _f=f, _fn=fn, _scal=scal, _ptrs=ptrs, _n=len(args)):
        key = tuple(("%g" % a[i] if i in _scal else ("0" if (i in _ptrs and not a[i]) else "*")) for i in range(min(_n, len(a))))
        counts[(_fn, key)] += 1
        return _f(*a)
    setattr(L, fn, wrapped)
engine.train_step(model, opt, *batch, crit)
torch.cuda.synchronize()
for fn, f in saved.items():
    setattr(L, fn, f)
print("%s W%dA%d qkr=%s, %d images: C-ABI calls of one eager step" % (name, bits, bits, qkr, B))
per = collections.Counter()
for (fn, key), n in counts.items():
    per[fn] += n
for fn, n in sorted(per.items(), key=lambda kv: -kv[1]):
    print("%5d  %s" % (n, fn))
    if only or n <= 64:
        for (f2, key), m in sorted(counts.items(), key=lambda kv: -kv[1]):
            if f2 == fn:
                print("        %4d x (%s)" % (m, ", ".join(key)))
