#!/usr/bin/env python3
"""Where does a two-process run on ONE shared GPU first stop being reproducible?  

Every repetition runs the same five eager training steps from the same state and leaves a TRACE of order-independent exact
checksums (the int32 bit patterns of a tensor summed in int64, no host sync while the steps run):

    (step, "loss")            the step's loss
    (step, "pre",  param)     a parameter's LOCAL gradient in its bucket slice, right before the bucket's all-reduce
    (step, "post", bucket)    the bucket after the collective's wait()
    (step, "grad", param)     p.grad as optimizer.step() sees it
    (step, "param", param)    the parameter after the step

A repetition whose trace differs from the first repetition's is reported with its FIRST differing entry.

    MODE=ranks   two ranks of one gloo group sharing GPU 0 (the failing case)
    MODE=solo    two INDEPENDENT processes sharing GPU 0, each with its own one-rank gloo group (or no wrapper: nodp)

    CFGS="base|sync|hand|bigbucket|nodefer|nodp"  ('+' combines), REPS=n, STEPS=5

    sync       torch.cuda.synchronize() before every all-reduce
    hand       hand-rolled all-reduce: synchronize, device -> host, dist.all_reduce on the CPU tensor, host -> device
    bigbucket  one bucket, reduced after the backward pass
    nodefer    no deferred dW groups, no deferred second-stage sums
    planes3    three bf16 planes (the exact backward GEMMs)
"""
import copy
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


class _Done:
    def wait(self):
        return True

    def is_completed(self):
        return True


def run_variant(torch, dist, engine, parallel, ops, Fn, base, batches, crit, flags, reps, steps, rank, world, tag):
    ck = lambda t: t.detach().contiguous().view(torch.int32).sum(dtype=torch.int64)      # noqa: E731
    ops._DBG_F16 = set() if "planes3" in flags else {"dx", "dw", "dqkx", "dxq"}
    Fn.SUM_DEFER = "nodefer" not in flags
    Fn.DW_GROUP = "nodefer" not in flags
    traces = []
    full0, details = None, {}
    want_full = os.environ.get("FULL") == "1"
    t0 = time.time()
    for rep in range(reps):
        model = copy.deepcopy(base).train()
        names = {id(p): n for n, p in model.named_parameters()}
        dp = None
        if "nodp" not in flags:
            dp = parallel.DataParallel(model, bucket_mb=1000.0 if "bigbucket" in flags else 1.0, force_sync=True)
        opt = engine.make_optimizer(model, lr=5e-4, weight_decay=0.05)
        keys, vals, step, full = [], [], [0], {}

        def rec(stage, name, t):
            keys.append((step[0], stage, name))
            vals.append(ck(t))
            if want_full and stage in ("pre", "grad"):
                full[(step[0], stage, name)] = t.detach().clone()

        if dp is not None:
            orig_ar, orig_fin = dp._all_reduce, dp.finish_gradient_sync

            def traced_ar(b, orig_ar=orig_ar, dp=dp):
                for v, p in zip(b.views, b.params):
                    rec("pre", names[id(p)], v)
                if "sync" in flags:
                    torch.cuda.synchronize()
                if "hand" in flags:
                    torch.cuda.synchronize()
                    h = b.flat.cpu()
                    dist.all_reduce(h)
                    h /= dp.world
                    b.flat.copy_(h)
                    return _Done()
                return orig_ar(b)

            def traced_fin(orig_fin=orig_fin, dp=dp):
                orig_fin()
                for i, b in enumerate(dp.buckets):
                    rec("post", "bucket%d" % i, b.flat)

            dp._all_reduce, dp.finish_gradient_sync = traced_ar, traced_fin
        orig_step = opt.step

        def traced_step(*a, orig_step=orig_step, model=model, **k):
            for n, p in model.named_parameters():
                if p.grad is not None:
                    rec("grad", n, p.grad)
            return orig_step(*a, **k)

        opt.step = traced_step
        for i in range(steps):
            step[0] = i
            loss = engine.train_step(model, opt, *batches[i % 2], crit, dp=dp)
            rec("loss", "", loss.detach().reshape(1))
            for n, p in model.named_parameters():
                rec("param", n, p)
        torch.cuda.synchronize()
        traces.append((keys, torch.stack(vals).cpu()))
        if want_full:
            if full0 is None:
                full0 = full
            elif not torch.equal(traces[-1][1], traces[0][1]) and keys == traces[0][0]:
                # every differing gradient of the FIRST differing step, in record (= arrival) order: how many elements, how far
                d = (traces[-1][1] != traces[0][1]).nonzero().reshape(-1).tolist()
                first_step = keys[d[0]][0]
                lines = []
                for j in d:
                    k = keys[j]
                    if k[0] != first_step or k not in full or k not in full0:
                        continue
                    a, b = full[k].double(), full0[k].double()
                    ne = (full[k] != full0[k])
                    idx = ne.reshape(-1).nonzero().reshape(-1)
                    lines.append("%s %-52s shape %-16s differing %d of %d, max |diff| %.3e, max |value| %.3e, first flat indices %s"
                                 % (k[:2], k[2], tuple(a.shape), int(ne.sum()), a.numel(), float((a - b).abs().max()),
                                    float(b.abs().max()), idx[:6].tolist()))
                details[rep] = lines
            del full
        if dp is not None:
            dp.release()
            del dp._all_reduce, dp.finish_gradient_sync
        if world > 1:
            dist.barrier()
    k0, v0 = traces[0]
    bad = []
    for r, (k, v) in enumerate(traces[1:], 1):
        if k != k0:
            bad.append((r, "trace KEYS differ (%d vs %d entries)" % (len(k), len(k0))))
            continue
        d = (v != v0).nonzero().reshape(-1)
        if d.numel():
            stages = {}
            for j in d.tolist():
                stages.setdefault(k[j][:2], 0)
                stages[k[j][:2]] += 1
            first = [k[j] for j in d[:4].tolist()]
            bad.append((r, "first %s  (%d entries differ; first stages %s)" % (first, d.numel(), list(stages.items())[:4])))
    print("%s rank %d cfg %-22s reps %d  %.1f s: %s" % (tag, rank, "+".join(sorted(flags)) or "base", reps, time.time() - t0,
                                                      "REPRODUCIBLE" if not bad else "%d of %d DIFFER" % (len(bad), reps - 1)), flush=True)
    for r, msg in bad[:6]:
        print("    rep %d: %s" % (r, msg), flush=True)
        for ln in details.get(r, [])[:40]:
            print("        " + ln, flush=True)
    return len(bad)


def worker(rank, world, port, mode):
    import torch
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port + (rank if mode == "solo" else 0))
    torch.cuda.set_device(0)
    if mode == "solo":
        dist.init_process_group("gloo", rank=0, world_size=1)
        gw = 1
    else:
        dist.init_process_group("gloo", rank=rank, world_size=world)
        gw = world
    from ofq_amd import engine, parallel, ops
    import ofq_amd.functional as Fn
    from ofq_amd.quantization.utils import KDLossSoftandHard
    torch.manual_seed(rank)
    base = engine.build_student("deit_tiny_distilled_patch16_224", 3, 3, qk_reparam=True, depth=2).cuda()
    g = torch.Generator(device="cuda").manual_seed(20 + rank)
    batches = [(torch.randn(4, 3, 224, 224, device="cuda", generator=g), torch.randint(0, 1000, (4,), device="cuda", generator=g),
                torch.randn(4, 1000, device="cuda", generator=g)) for _ in range(2)]
    engine.setup_alpha(base, batches[0][0])
    crit = KDLossSoftandHard()
    reps, steps = int(os.environ.get("REPS", "40")), int(os.environ.get("STEPS", "5"))
    total = 0
    for cfg in os.environ.get("CFGS", "base").split("|"):
        flags = set(f for f in cfg.split("+") if f and f != "base")
        total += run_variant(torch, dist, engine, parallel, ops, Fn, base, batches, crit, flags, reps, steps, rank, gw, mode)
    dist.destroy_process_group()


if __name__ == "__main__":
    import socket
    import torch.multiprocessing as mp
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    mode = os.environ.get("MODE", "ranks")
    mp.spawn(worker, args=(2, port, mode), nprocs=int(os.environ.get("PROCS", "2")), join=True)
