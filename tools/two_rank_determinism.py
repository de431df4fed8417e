#!/usr/bin/env python3
"""Two ranks SHARING one GPU over gloo (world 2): the eager DataParallel step, 5 steps, repeated REPS times from the same state for each
of several ops._DBG_F16 configurations; reports whether the repetitions agree bit for bit (both ranks)."""
import copy, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def worker(rank, world, port):
    import torch
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from ofq_amd import engine, parallel, ops
    from ofq_amd.quantization.utils import KDLossSoftandHard
    torch.manual_seed(rank)
    base = engine.build_student("deit_tiny_distilled_patch16_224", 3, 3, qk_reparam=True, depth=2).cuda()
    g = torch.Generator(device="cuda").manual_seed(20 + rank)
    batches = [(torch.randn(4, 3, 224, 224, device="cuda", generator=g), torch.randint(0, 1000, (4,), device="cuda", generator=g),
                torch.randn(4, 1000, device="cuda", generator=g)) for _ in range(2)]
    engine.setup_alpha(base, batches[0][0])
    crit = KDLossSoftandHard()
    reps = int(os.environ.get("REPS", "6"))
    for cfg in os.environ.get("CFGS", "dx,dw,dqkx,dxq|dx,dw|dqkx,dxq|none").split("|"):
        cfg, _, extra = cfg.partition("+")
        ops._DBG_F16 = set(cfg.split(","))
        import ofq_amd.functional as Fn
        Fn.SUM_DEFER = "nosum" not in extra
        Fn.DW_GROUP = "nodwgroup" not in extra
        bucket_mb = 1000.0 if "bigbucket" in extra else 1.0
        cfg = cfg + "+" + extra
        runs = []
        for rep in range(reps):
            model = copy.deepcopy(base).train()
            dp = parallel.DataParallel(model, bucket_mb=bucket_mb)
            if "noslot" in extra:
                parallel._GRAD_SLOTS.clear()
            opt = engine.make_optimizer(model, lr=5e-4, weight_decay=0.05)
            losses = [float(engine.train_step(model, opt, *batches[i % 2], crit, dp=dp).detach()) for i in range(5)]
            torch.cuda.synchronize()
            runs.append((losses, torch.cat([p.detach().reshape(-1) for p in model.parameters()]).clone()))
            dp.release()
            dist.barrier()
        same = [r[0] == runs[0][0] and torch.equal(r[1], runs[0][1]) for r in runs]
        print("rank %d cfg %-22s: %s  %s" % (rank, cfg, "DETERMINISTIC" if all(same) else "DIFFERS %s" % same,
                                             ["%.6f" % r[0][-1] for r in runs]), flush=True)
    dist.destroy_process_group()


if __name__ == "__main__":
    import socket
    import torch.multiprocessing as mp
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    mp.spawn(worker, args=(2, port), nprocs=2, join=True)
