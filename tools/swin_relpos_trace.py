#!/usr/bin/env python3
"""Where does the relative-position-bias gradient of Swin-T's first block start to differ between replays of the captured step?
(tools/step_soak_determinism.py MODEL=swin_t MODE=graph: that one gradient took 16 values over 500 replays, every other gradient and
200 eager steps were identical; tools/relpos_determinism_probe.py: the two ops of its backward are reproducible on their own.)

The captured step is instrumented from outside: functional._addend_grad and swin._RelPosBiasFn.backward are wrapped so that, during
the capture, clones of their inputs and outputs are kept (clone kernels inside the graph, so every replay refreshes them); after
each replay the clones' exact checksums are compared with the first replay's."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from ofq_amd import engine, swin
import ofq_amd.functional as Fn
from ofq_amd.quantization.utils import KDLossSoftandHard

STEPS = int(os.environ.get("STEPS", "300"))
torch.manual_seed(0)
model = engine.build_student("swin_t", 3, 3, qk_reparam=True).cuda()
g = torch.Generator(device="cuda").manual_seed(11)
batch = (torch.randn(128, 3, 224, 224, device="cuda", generator=g), torch.randint(0, 1000, (128,), device="cuda", generator=g),
         torch.randn(128, 1000, device="cuda", generator=g))
engine.setup_alpha(model, batch[0][:16])
model.train()
opt = engine.make_optimizer(model, lr=0.0, weight_decay=0.0)
kept = []          # (label, tensor clone)
real_ag, real_rp = Fn._addend_grad, swin._RelPosBiasFn.backward
calls = {"ag": 0, "rp": 0}


def ag(dS, addend, alpha):
    out = real_ag(dS, addend, alpha)
    if torch.cuda.is_current_stream_capturing():
        calls["ag"] += 1
        if addend.shape[0] <= 24:                       # the un-shifted blocks: slabs = heads
            kept.append(("addend_grad#%d in dS%s" % (calls["ag"], tuple(dS.shape)), dS.clone()))
            kept.append(("addend_grad#%d out%s" % (calls["ag"], tuple(out.shape)), out.clone()))
    return out


def rp(ctx, gr):
    out = real_rp(ctx, gr)
    if torch.cuda.is_current_stream_capturing():
        calls["rp"] += 1
        kept.append(("relpos_bwd#%d in%s" % (calls["rp"], tuple(gr.shape)), gr.clone()))
        kept.append(("relpos_bwd#%d out%s" % (calls["rp"], tuple(out[0].shape)), out[0].clone()))
    return out


Fn._addend_grad = ag
swin._RelPosBiasFn.backward = staticmethod(rp)
step = engine.GraphedTrainStep(model, opt, KDLossSoftandHard(), alias_inputs=True)
for _ in range(4):
    step(*batch)
torch.cuda.synchronize()
names = [n for n, p in model.named_parameters() if "relative_position_bias_table" in n]
params = dict(model.named_parameters())


def sums():
    v = [t.contiguous().view(torch.int32).sum(dtype=torch.int64) for _, t in kept]
    v += [params[n].grad.contiguous().view(torch.int32).sum(dtype=torch.int64) for n in names]
    return torch.stack(v)


rows = []
for i in range(STEPS):
    step(*batch)
    rows.append(sums())
torch.cuda.synchronize()
r = torch.stack(rows).cpu()
labels = [l for l, _ in kept] + ["grad " + n for n in names]
print("%d kept tensors, %d replays" % (len(kept), STEPS))
for j, l in enumerate(labels):
    col = r[:, j]
    nd = len(torch.unique(col))
    if nd > 1 or "relpos_bwd#" in l or "grad " in l:
        print("%-70s %d distinct values over the replays" % (l, nd))
