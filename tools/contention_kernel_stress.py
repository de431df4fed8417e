#!/usr/bin/env python3
"""Which KERNEL stops being reproducible when another process shares the GPU?  (after tools/two_rank_trace.py
had named the first differing op.)  Two independent processes; each runs one tiny training step (DeiT-T width, depth 2, W3A3 QKR,
4 images: the configuration of the two-rank test), captures the arguments of the first call of every op in OPS during that step,
and then launches each captured call REPS times on the SAME inputs, comparing every tensor it returns with the first launch's,
bit for bit, while the other process does the same (mutual contention).  Prints, per op and output, the number of launches that
differed and the largest difference.

    OPS="qattn_dp_softmax_bwd,qattn_scores_softmax,..."   REPS=3000   PROCS=2 (1: no contention, the control)
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def worker(rank, nprocs):
    import torch
    torch.cuda.set_device(0)
    from ofq_amd import engine, ops
    from ofq_amd.quantization.utils import KDLossSoftandHard
    torch.manual_seed(rank)
    model = engine.build_student("deit_tiny_distilled_patch16_224", 3, 3, qk_reparam=True, depth=2).cuda()
    g = torch.Generator(device="cuda").manual_seed(20 + rank)
    batch = (torch.randn(4, 3, 224, 224, device="cuda", generator=g), torch.randint(0, 1000, (4,), device="cuda", generator=g),
             torch.randn(4, 1000, device="cuda", generator=g))
    engine.setup_alpha(model, batch[0])
    model.train()
    opt = engine.make_optimizer(model, lr=5e-4, weight_decay=0.05)
    names = os.environ.get("OPS", "qattn_dp_softmax_bwd,qattn_scores_softmax,qattn_dqkx,qattn_dxq,qattn_dv,qattn_pv,lsq_bwd,"
                                  "qgemm_i8_lsq_bwd,layernorm_lsq_bwd,softmax_lsq_bwd").split(",")
    captured, saved = {}, {}

    def clone(x):
        if isinstance(x, torch.Tensor):
            return x.detach().clone()
        if isinstance(x, (list, tuple)):
            return type(x)(clone(y) for y in x)
        if isinstance(x, dict):
            return {k: clone(v) for k, v in x.items()}
        return x

    for n in names:
        fn = getattr(ops, n, None)
        if fn is None:
            continue
        saved[n] = fn

        def wrapped(*a, _n=n, _fn=fn, **k):
            if _n not in captured:
                captured[_n] = (clone(a), clone(k))
            return _fn(*a, **k)
        setattr(ops, n, wrapped)
    engine.train_step(model, opt, *batch, KDLossSoftandHard())
    torch.cuda.synchronize()
    for n, fn in saved.items():
        setattr(ops, n, fn)
    reps = int(os.environ.get("REPS", "3000"))

    def flat(r, out):
        if isinstance(r, torch.Tensor):
            out.append(r)
        elif isinstance(r, (list, tuple)):
            for y in r:
                flat(y, out)
        return out

    for n in names:
        if n not in captured:
            print("proc %d %-24s not called in the step" % (rank, n), flush=True)
            continue
        a, k = captured[n]
        fn = saved[n]
        first = [t.clone() for t in flat(fn(*clone(a), **clone(k)), [])]
        torch.cuda.synchronize()
        bad = [torch.zeros((), dtype=torch.int64, device="cuda") for _ in first]
        worst = [torch.zeros((), dtype=torch.float64, device="cuda") for _ in first]
        inplace = n in ("qattn_dxq", "softmax_lsq_bwd", "lsq_bwd", "layernorm_lsq_bwd")     # (may write into an argument)
        detail = os.environ.get("DETAIL") == "1"
        shown = 0
        for it in range(reps):
            outs = flat(fn(*clone(a), **clone(k)) if inplace else fn(*a, **k), [])
            for j, (o, f) in enumerate(zip(outs, first)):
                ne = (o != f) & ~(torch.isnan(o) & torch.isnan(f)) if o.dtype.is_floating_point else (o != f)
                bad[j] += ne.any().long()
                if o.dtype.is_floating_point:
                    worst[j] = torch.maximum(worst[j], (o.double() - f.double()).abs().max())
                if detail and shown < 12 and o.dim() == 1 and bool(ne.any()):       # (a host sync per launch: DETAIL runs only)
                    shown += 1
                    for r_ in ne.nonzero().reshape(-1)[:4].tolist():
                        near = {d: float(f[r_ + d]) for d in (-48, -32, -16, 16, 32, 48) if 0 <= r_ + d < f.numel()}
                        hit = [d for d, v in near.items() if v == float(o[r_])]
                        print("proc %d %s launch %d out%d[%d] = %.9g instead of %.9g; equals the correct value of element r%+d: %s"
                              % (rank, n, it, j, r_, float(o[r_]), float(f[r_]), hit[0] if hit else 0, bool(hit)), flush=True)
        torch.cuda.synchronize()
        print("proc %d %-24s %d launches: %s" % (rank, n, reps, "  ".join(
            "out%d%s differed %d x (max |diff| %.2e of %.2e)" % (j, tuple(f.shape), int(b), float(w), float(f.double().abs().max()) if f.dtype.is_floating_point else 0.0)
            for j, (f, b, w) in enumerate(zip(first, bad, worst)))), flush=True)


if __name__ == "__main__":
    import torch.multiprocessing as mp
    n = int(os.environ.get("PROCS", "2"))
    mp.spawn(worker, args=(n,), nprocs=n, join=True)
