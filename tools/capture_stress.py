#!/usr/bin/env python3
"""Capture / re-capture of the training step next to a LIVE process group (one rank over RCCL, --force-dp --sync-statsq style):
N re-captures in a row, each right behind two eager steps whose collectives c10d's watchdog thread may not have retired yet.
Variants (each in its own child process: the failure mode is the watchdog thread ending the process):
  global        capture_error_mode "global", no drain   (round 3's capture: what crashed in profiles/r04_configs.txt)
  thread_local  capture_error_mode "thread_local", no drain
  product       thread_local + DataParallel.drain_collectives() (engine.GraphedTrainStep's default), no sleep anywhere
  python tools/capture_stress.py [N]      -> one line per variant: ok / exit code + the last line of its stderr"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def child(variant, n, graph_mode):
    import torch
    import torch.distributed as dist
    from ofq_amd import engine, parallel
    from ofq_amd.quantization.utils import KDLossSoftandHard
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29547")
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    if variant != "product":
        engine.GraphedTrainStep.drain_before_capture = False
        engine.GraphedTrainStep.capture_error_mode_dp = "global" if variant == "global" else "thread_local"
    torch.manual_seed(0)
    big = os.environ.get("STRESS_BIG") == "1"           # the bench.py workload: DeiT-S W2A2 QKR, 128 images, 24 MB buckets
    B = 128 if big else 8
    if big:
        model = engine.build_student("deit_small_distilled_patch16_224", 2, 2, qk_reparam=True).to(dev)
    else:
        model = engine.build_student("deit_tiny_distilled_patch16_224", 3, 3, qk_reparam=True, depth=2).to(dev)
    g = torch.Generator(device=dev).manual_seed(1)
    imgs = torch.randn(B, 3, 224, 224, device=dev, generator=g)
    tgt = torch.randint(0, 1000, (B,), device=dev, generator=g)
    soft = torch.randn(B, 1000, device=dev, generator=g)
    engine.setup_alpha(model, imgs)
    model.train()
    dp = parallel.DataParallel(model, bucket_mb=24.0 if big else 1.0, force_sync=True, sync_statsq=True)
    opt = engine.make_optimizer(model)
    crit = KDLossSoftandHard()
    gs = engine.GraphedTrainStep(model, opt, crit, dp=dp, warmup=2, mode=graph_mode)
    for i in range(n):
        # two eager steps (collectives in flight on RCCL's stream), then a capture right behind them, then two replays
        with torch.cuda.stream(gs.stream):           # (the stream the capture uses: autograd nodes remember their stream)
            engine.train_step(model, opt, imgs, tgt, soft, crit, dp=dp)
            engine.train_step(model, opt, imgs, tgt, soft, crit, dp=dp)
        torch.cuda.current_stream().wait_stream(gs.stream)
        gs.graph = None
        gs.calls = max(gs.calls, gs.warmup)
        gs(imgs, tgt, soft)
        gs(imgs, tgt, soft)
    torch.cuda.synchronize()
    assert gs.captures == n, gs.captures
    print("captures", gs.captures, "loss", float(gs.loss))
    dist.destroy_process_group()


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "--child":
        child(sys.argv[2], int(sys.argv[3]), sys.argv[4])
        sys.exit(0)
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 20
    for graph_mode in ("segmented", "split"):
        for variant in ("product", "thread_local", "global"):
            r = subprocess.run([sys.executable, os.path.abspath(__file__), "--child", variant, str(n), graph_mode], capture_output=True,
                               text=True, timeout=900)
            tail = [l for l in (r.stderr or "").strip().splitlines() if l.strip() and "amdgpu.ids" not in l][-1:] if r.returncode else []
            what = [l for l in (r.stderr or "").splitlines() if "HIP error" in l or "Error" in l][:2]
            print("%-9s %-12s %d re-captures: %s %s" % (graph_mode, variant, n, "ok (" + r.stdout.strip().splitlines()[-1] + ")" if r.returncode == 0
                                                  else "EXIT %d" % r.returncode, " | ".join(what + tail)[:300]), flush=True)
