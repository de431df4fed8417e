#!/usr/bin/env python3
"""Which op of the relative-position-bias gradient differs between replays of a captured Swin step?  (Round 6: the full-size Swin-T
soak found ONE gradient, features.1.0.attn.relative_position_bias_table, taking 16 different values over 500 replays while 200 eager
steps were identical.)  Its backward is (a) the sum of dS over the windows that share a bias slab (swin stage 1, un-shifted: 8192
windows, 3 heads) and (b) a one-hot product that scatters the (N*N, heads) result into the (169, heads) table.  Both as they were
(torch.sum / torch.matmul) and as they are now (ops.colsum / a padded gather + sum), REPS replays each from one captured graph."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from ofq_amd import ops

REPS = int(os.environ.get("REPS", "300"))
g = torch.Generator(device="cuda").manual_seed(0)
G, P, N, Np, T = 8192, 3, 49, 64, 169
dS = torch.randn(G, P, N, Np, device="cuda", generator=g) * 1e-3
idx = torch.randint(0, T, (N * N,), device="cuda", generator=g)
oh = torch.zeros(T, N * N, device="cuda")
oh[idx, torch.arange(N * N, device="cuda")] = 1.0


def run(name, fn):
    out = fn()
    torch.cuda.synchronize()
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr, stream=s):
        out = fn()
    variants = {}
    for i in range(REPS):
        gr.replay()
        torch.cuda.synchronize()
        k = int(out.view(torch.int32).sum(dtype=torch.int64))
        variants.setdefault(k, 0)
        variants[k] += 1
    eager = {}
    for i in range(REPS):
        o = fn()
        k = int(o.view(torch.int32).sum(dtype=torch.int64))
        eager.setdefault(k, 0)
        eager[k] += 1
    print("%-44s replayed: %d distinct results %s   eager: %d distinct %s" % (name, len(variants), sorted(variants.values(), reverse=True)[:6],
                                                                            len(eager), sorted(eager.values(), reverse=True)[:6]), flush=True)
    return out


summed = run("torch.sum over 8192 windows", lambda: dS.sum(0))
g2 = summed[..., :N].permute(1, 2, 0).reshape(N * N, P).contiguous()
run("torch.matmul one-hot (169 x 2401) @ (2401 x 3)", lambda: oh @ g2)
run("ops.colsum over 8192 windows", lambda: ops.colsum(dS.view(G, P * N * Np)))
cnt = torch.bincount(idx, minlength=T)
pos = torch.full((T, int(cnt.max())), N * N, dtype=torch.long, device="cuda")
order = torch.argsort(idx, stable=True)
start = torch.cumsum(cnt, 0) - cnt
for t in range(T):
    pos[t, :int(cnt[t])] = order[int(start[t]):int(start[t]) + int(cnt[t])]
flat = pos.view(-1)


def gather():
    ext = torch.cat([g2, torch.zeros(1, P, device="cuda")])
    return ext.index_select(0, flat).view(T, -1, P).sum(1)


o2 = run("padded gather + sum(1)", gather)
print("gather vs matmul: max |diff| %.3e of %.3e" % (float((o2 - oh @ g2).abs().max()), float((oh @ g2).abs().max())))
