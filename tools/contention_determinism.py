#!/usr/bin/env python3
"""Is the eager training step deterministic while ANOTHER process keeps the same GPU busy?  Two independent processes (no process
group between them) each run the same 6 steps three times from the same state and compare losses / parameters bit for bit.
DP=1: with the DataParallel wrapper over a one-rank gloo group (hooks, buckets, deferred-dW flushes inside the hooks)."""
import copy, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def child(idx):
    import torch
    from ofq_amd import engine, parallel
    from ofq_amd.quantization.utils import KDLossSoftandHard
    torch.cuda.set_device(0)
    dp_on = os.environ.get("DP") == "1"
    if dp_on:
        import torch.distributed as dist
        os.environ["MASTER_ADDR"] = "127.0.0.1"
        os.environ["MASTER_PORT"] = str(29600 + idx)
        dist.init_process_group("gloo", rank=0, world_size=1)
    torch.manual_seed(idx)
    base = engine.build_student("deit_tiny_distilled_patch16_224", 3, 3, qk_reparam=True, depth=2).cuda()
    g = torch.Generator(device="cuda").manual_seed(10 + idx)
    batches = [(torch.randn(4, 3, 224, 224, device="cuda", generator=g), torch.randint(0, 1000, (4,), device="cuda", generator=g),
                torch.randn(4, 1000, device="cuda", generator=g)) for _ in range(2)]
    engine.setup_alpha(base, batches[0][0])
    crit = KDLossSoftandHard()
    runs = []
    for rep in range(int(os.environ.get("REPS", "4"))):
        model = copy.deepcopy(base).train()
        dp = parallel.DataParallel(model, bucket_mb=1.0, force_sync=True) if dp_on else None
        opt = engine.make_optimizer(model, lr=5e-4, weight_decay=0.05)
        losses = []
        for i in range(6):
            losses.append(float(engine.train_step(model, opt, *batches[i % 2], crit, dp=dp).detach()))
        torch.cuda.synchronize()
        runs.append((losses, torch.cat([p.detach().reshape(-1) for p in model.parameters()]).clone()))
        if dp is not None:
            dp.release()
    same = all(r[0] == runs[0][0] and torch.equal(r[1], runs[0][1]) for r in runs[1:])
    print("proc %d dp=%s planes=%s f16=%s: %s %s" % (idx, dp_on, os.environ.get("OFQ_GRAD_PLANES", "2"), os.environ.get("OFQ_DEBUG_F16", "all"),
                                                   "DETERMINISTIC" if same else "DIFFERS", [r[0][-1] for r in runs]), flush=True)


if __name__ == "__main__":
    if len(sys.argv) > 2 and sys.argv[1] == "--child":
        child(int(sys.argv[2]))
        sys.exit(0)
    n = int(os.environ.get("PROCS", "2"))
    ps = [subprocess.Popen([sys.executable, os.path.abspath(__file__), "--child", str(i)]) for i in range(n)]
    sys.exit(max(p.wait() for p in ps))
