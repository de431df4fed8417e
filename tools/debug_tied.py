#!/usr/bin/env python3
"""A block used twice in one step: which switch makes engine's step differ from a plain backward pass?"""
import os, sys, copy
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from ofq_amd import engine, ops
import ofq_amd.functional as Fn
from ofq_amd.quantization.utils import KDLossSoftandHard
from util import rel_err

torch.manual_seed(0)
base = engine.build_student("deit_tiny_distilled_patch16_224", 3, 3, qk_reparam=True, depth=2).cuda()
g = torch.Generator(device="cuda").manual_seed(5)
b0 = (torch.randn(4, 3, 224, 224, device="cuda", generator=g), torch.randint(0, 1000, (4,), device="cuda", generator=g),
      torch.randn(4, 1000, device="cuda", generator=g))
engine.setup_alpha(base, b0[0])
tied = os.environ.get("TIED", "1") == "1"
if tied:
    base.blocks[1] = base.blocks[0]
loss_fn = KDLossSoftandHard()


def grads(mode):
    model = copy.deepcopy(base).train()
    if mode == "plain":
        out, _ = model(b0[0])
        loss_fn(out, b0[1], b0[2]).backward()
    else:
        opt = engine.make_optimizer(model, lr=0.0, weight_decay=0.0)
        engine.train_step(model, opt, *b0)
    return {n: p.grad.detach().clone() for n, p in model.named_parameters() if p.grad is not None}


ref = grads("plain")
for name, setter in [("default", lambda: None), ("default again", lambda: None),
                     ("SUM_DEFER off", lambda: setattr(Fn, "SUM_DEFER", False)),
                     ("DW_GROUP off", lambda: setattr(Fn, "DW_GROUP", False)),
                     ("code cache off", lambda: setattr(engine, "WEIGHT_CODE_CACHE", False)),
                     ("concat off", lambda: setattr(ops, "NT_CONCAT", False))]:
    setter()
    got = grads("step")
    bad = [(n, rel_err(got[n].cpu(), ref[n].cpu())) for n in ref if rel_err(got[n].cpu(), ref[n].cpu()) > 1e-5]
    print("%-16s: %d of %d parameters off: %s" % (name, len(bad), len(ref), ", ".join("%s %.1e" % b for b in bad[:8])), flush=True)
