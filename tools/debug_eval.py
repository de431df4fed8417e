#!/usr/bin/env python3
"""Eval / train forward of a full-depth DeiT-T W3A3 QKR model against the oracle, fresh and after a few steps."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import torch
import ofq_oracle as O
from ofq_amd import engine
from ofq_amd.quantization.utils import KDLossSoftandHard

def cmp(model, x, tag, depth, bits):
    sd = {k: v.detach().cpu() for k, v in model.state_dict().items()}
    cfg = dict(depth=depth, num_heads=3, patch=16, wbits=bits, abits=bits, qkr=True)
    for training in (False, True):
        model.train(training)
        with torch.no_grad():
            out, _ = model(x)
            ref = O.deit_forward(x.cpu(), sd, cfg, training=training)
        if training:
            out, ref = out[0], ref[0]
        e = float((out.cpu().double() - ref.double()).norm() / ref.double().norm())
        print("%s depth=%d bits=%d training=%s: rel L2 %.3e  (|ref| %.3f)" % (tag, depth, bits, training, e, float(ref.norm())))

for depth, bits in ((2, 3), (12, 3), (12, 2), (12, 4)):
    torch.manual_seed(0)
    model = engine.build_student("deit_tiny_distilled_patch16_224", bits, bits, qk_reparam=True, depth=depth).cuda()
    x = torch.randn(4, 3, 224, 224, device="cuda")
    tgt = torch.randint(0, 1000, (4,), device="cuda")
    soft = torch.randn(4, 1000, device="cuda")
    engine.setup_alpha(model, x)
    cmp(model, x, "fresh", depth, bits)
    model.train()
    opt = engine.make_optimizer(model, lr=5e-4)
    for _ in range(4):
        engine.train_step(model, opt, x, tgt, soft, KDLossSoftandHard())
    cmp(model, x, "after 4 steps", depth, bits)
