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


_SKIP_OPS = {"lib", "workspace", "placeholder", "num_cus", "amax_begin", "amax_end", "amax_word", "amax_of", "amax_for", "amax_out",
             "tag_amax", "sum_flush", "sum_pending", "sum_drop", "nt_sk_poll", "nt_sk_poison", "nt_sk_error", "nt_sk_pays",
             "nt_concat_ok", "tn_tiles", "tn_groupable", "nt_class", "scores_amax", "deferred_sums", "lsq_eff_scale"}


def wrap_ops(torch, ops, rec_op):
    """OPS=1: every wrapper in ofq_amd.ops records order-independent checksums of its tensor arguments and results, in call order:
    the first call whose inputs equal the first repetition's and whose outputs do not is the kernel that is not reproducible."""
    import types
    saved = {}

    def tensors(x, out, depth=0):
        if isinstance(x, torch.Tensor):
            if x.is_cuda and x.numel() > 0 and x.dtype in (torch.float32, torch.int8, torch.float16, torch.bfloat16, torch.int32):
                out.append(x)
        elif isinstance(x, (list, tuple)) and depth < 3:
            for y in x:
                tensors(y, out, depth + 1)
        elif isinstance(x, dict) and depth < 3:
            for y in x.values():
                tensors(y, out, depth + 1)

    def make(name, fn):
        def wrapped(*a, **k):
            ins = []
            tensors(a, ins)
            tensors(k, ins)
            for i, t in enumerate(ins):
                rec_op(name, "in%d" % i, t)
            r = fn(*a, **k)
            outs = []
            tensors(r, outs)
            if name == os.environ.get("DUMP_OP") and DUMPS is not None:
                DUMPS.append(([t.detach().clone() for t in ins], [t.detach().clone() for t in outs], a, k, fn))
            for i, t in enumerate(outs):
                rec_op(name, "out%d" % i, t)
            for key in ("out",):
                if isinstance(k.get(key), torch.Tensor):
                    rec_op(name, "kw_" + key, k[key])
            return r
        return wrapped

    for name, fn in list(vars(ops).items()):
        if isinstance(fn, types.FunctionType) and not name.startswith("_") and name not in _SKIP_OPS and fn.__module__ == ops.__name__:
            saved[name] = fn
            setattr(ops, name, make(name, fn))
    return saved


DUMPS = None       # DUMP_OP=<op>: (cloned inputs, cloned outputs, args, kwargs, fn) of every call of that op in the running repetition


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

        saved_ops = None
        global DUMPS
        DUMPS = [] if os.environ.get("DUMP_OP") else None
        if os.environ.get("OPS") == "1":
            calls = [0]

            in_bwd = [False]
            orig_bb, orig_end = Fn.begin_backward, Fn.assert_step_queues_empty

            def bb(orig_bb=orig_bb):
                in_bwd[0] = True
                return orig_bb()

            def be(orig_end=orig_end):
                in_bwd[0] = False
                return orig_end()
            Fn.begin_backward, Fn.assert_step_queues_empty = bb, be

            def rec_op(name, what, t):
                # backward pass only; the primary output (written by the launch itself: the other outputs are parameter-gradient
                # reductions that may still wait in the deferred queues) and the first two inputs
                full_in = name in ("qattn_dp_softmax_bwd",)          # (every input of the op the first runs pointed at)
                fwd_ops = ("qattn_prep", "qattn_scores_softmax", "qattn_pv")     # ... and, in the FORWARD pass, its producers
                if not in_bwd[0]:
                    if name not in fwd_ops:
                        return
                elif name in ("gemm", "absmax") or (what not in ("out0", "kw_out", "in0", "in1") and not
                                                    (full_in and what.startswith("in"))):
                    return
                calls[0] += 1
                keys.append((step[0], "op", "%05d %s %s %s" % (calls[0], name, what, tuple(t.shape))))
                tt = t.detach()
                if tt.dtype in (torch.float16, torch.bfloat16):
                    tt = tt.contiguous().view(torch.int16).to(torch.int32)
                elif tt.dtype == torch.int8:
                    tt = tt.to(torch.int32)
                vals.append(tt.contiguous().view(torch.int32).sum(dtype=torch.int64))
            saved_ops = wrap_ops(torch, ops, rec_op)
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
        if saved_ops is not None:
            for name, fn in saved_ops.items():
                setattr(ops, name, fn)
            Fn.begin_backward, Fn.assert_step_queues_empty = orig_bb, orig_end
        traces.append((keys, torch.stack(vals).cpu()))
        if DUMPS is not None:
            if rep == 0:
                dumps0 = DUMPS
            elif not torch.equal(traces[-1][1], traces[0][1]):
                # the dumped op's calls of this repetition against the first repetition's, and against a fresh launch on the
                # cloned inputs now (nothing else of this process in flight)
                for ci, ((ins, outs, a, k, fn), (ins0, outs0, _, _, _)) in enumerate(zip(DUMPS, dumps0)):
                    same_in = all(torch.equal(x, y) for x, y in zip(ins, ins0))
                    for oi, (o, o0) in enumerate(zip(outs, outs0)):
                        if not torch.equal(o, o0):
                            idx = (o != o0).reshape(-1).nonzero().reshape(-1)
                            again = []
                            tensors_again = saved_ops[os.environ["DUMP_OP"]](*a, **k)
                            flat = list(tensors_again) if isinstance(tensors_again, (tuple, list)) else [tensors_again]
                            ag = flat[oi]
                            print("    rep %d rank %d: %s call %d out%d differs in %d of %d elements (inputs bit-equal: %s); first indices %s\n"
                                  "        this rep %s\n        first rep %s\n        fresh launch on the same arguments now: equals this rep %s, equals the first rep %s"
                                  % (rep, rank, os.environ["DUMP_OP"], ci, oi, idx.numel(), o.numel(), same_in, idx[:8].tolist(),
                                     o.reshape(-1)[idx[:8]].tolist(), o0.reshape(-1)[idx[:8]].tolist(), bool(torch.equal(ag, o)),
                                     bool(torch.equal(ag, o0))), flush=True)
                            if os.environ["DUMP_OP"] == "qattn_prep" and oi == 1:
                                # tq[r] = qcodes[r, :] . bax: which of the two values is the dot product of the inputs as they are
                                # in memory, and which single term explains the other one?
                                qc, bax = ins[2].reshape(o.numel(), -1).double(), ins[3].reshape(-1).double()
                                qc0, bax0 = ins0[2].reshape(o.numel(), -1).double(), ins0[3].reshape(-1).double()
                                for r_ in idx[:4].tolist():
                                    ref = float((qc[r_] * bax).sum())
                                    wrong, right = float(o.reshape(-1)[r_]), float(o0.reshape(-1)[r_])
                                    d_this, d_first = wrong - ref, right - ref
                                    cand = []
                                    for val, dlt in (("this", d_this), ("first", d_first)):
                                        if abs(dlt) > 1e-6:
                                            ratio = dlt / bax
                                            near = ((ratio - ratio.round()).abs() < 2e-3) & (ratio.round().abs() >= 1) & (ratio.round().abs() <= 16)
                                            cand.append((val, [(int(kk), int(ratio[kk].round()), int(qc[r_, kk])) for kk in near.nonzero().reshape(-1)[:6].tolist()]))
                                    o0f = o0.reshape(-1)
                                    hits = [d_ for d_ in (-48, -32, -16, 16, 32, 48) if 0 <= r_ + d_ < o0f.numel() and float(o0f[r_ + d_]) == wrong]
                                    print("        row %d (row %% 64 = %d): the wrong value IS the correct value of row r%+d: %s" % (r_, r_ % 64, hits[0] if hits else 0, bool(hits)), flush=True)
                                    print("        row %d: fp64 dot of the inputs %.9f; this rep - dot %.3e, first rep - dot %.3e; single-term explanations "
                                          "(k, code delta, code): %s; inputs of this call equal the first rep's: codes %s bax %s"
                                          % (r_, ref, d_this, d_first, cand, bool(torch.equal(qc, qc0)), bool(torch.equal(bax, bax0))), flush=True)
                            break
                    else:
                        continue
                    break
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
            first = [k[j] for j in d[:(40 if os.environ.get("OPS") == "1" else 4)].tolist()]
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
