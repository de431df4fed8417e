#!/usr/bin/env python3
"""bench.py — images/sec of one OFQ QAT training step on MI355X (BASELINE.json's metric).

    python bench.py --gpus N --steps K --warmup W          (N > 1: one rank per GPU -- started by torch.distributed.run,
                                                            or, without a launcher, by this file itself as N children)

Workload (config C3 of BASELINE.json / SURVEY.md §8d): DeiT-S (C384, H6, 12 blocks, 198 tokens) W2A2 with
QK-reparameterisation, 128 images per GPU (global batch 128*N, weak scaling), fp32, synthetic ImageNet-shaped
batches (randn images, random labels, random teacher logits), random-init weights.  One step = student
forward + KDLossSoftandHard + backward (+ bucketed RCCL all-reduce for N > 1) + AdamW step; the device side of the step
is captured once in a hipGraph (engine.GraphedTrainStep) and replayed -- every kernel of every timed step runs.

`dtype` / `config.grad_planes` say which arithmetic the backward code GEMMs ran on (default: the gradient operand as two fp16 planes of
the power-of-two-scaled tensor; OFQ_GRAD_PLANES=3: three bf16 planes, the exact fp32 product); the default line also
carries `exact_fp32_backward` (the same 20 steps with OFQ_GRAD_PLANES=3) and `recipe_step` (the fp32 KD teacher's forward inside the
step), each measured by a child process after the headline's timed region.

Prints ONE JSON line on rank 0, with two extra objects:
  roofline      — the dominant matrix-core kernel class by time (at 128 images: qgemm_bf16s_nt_wide_kernel, the dX GEMM of
                  the linear layers): algorithmic 2*M*N*K per launch / HIP-event time per launch, both
                  measured live over a second pass of the timed steps, against the 2.5 PFLOP/s dense bf16 MFMA peak
                  (`frac`); every algorithmic FMA of the split kernels is three bf16 MFMA FMAs, so the matrix pipe is busy
                  3x that fraction (`mfma_pipe_frac`); `mfma_util_pmc` / `traffic` come from the committed rocprofv3 PMC
                  passes of the same kernel sources (profiles/r06_traffic.json)
  cpu_baseline  — oracle/ofq_oracle.py (eager torch-CPU restatement of the reference path) timed on this box's
                  host cores on a bounded sample (DeiT-S W2A2 QKR, batch 8, a few steps), rank 0 at N = 1 only
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

# dense matrix-core peaks, /opt/skills/guides/MI355X_MICROARCH.md ("Chip-level parameters" / "Matrix cores (MFMA)")
PEAKS = {"gemm_f32": 157.3,        # Peak FP32 (matrix)
         "qgemm_bf16s": 2500.0,    # Peak BF16 MFMA dense; the fp32-exact product issues 3 bf16 MFMAs per algorithmic FMA
         "qgemm_i8": 5000.0,       # I8 runs at 2x the bf16 rate (2xK); measured ceiling in the guide: 3944-4404 TOPS
         "qattn_scores_softmax": 5000.0,   # (fused int8 GEMM + softmax kernels have their own timer classes: they are VALU-bound)
         "qattn_dp_softmax_bwd": 2500.0}   # (fused dP GEMM + softmax backward: VALU / HBM-bound)
PROFILE_JSON = "r06_traffic.json"  # profiles/: per-kernel HBM bytes and MfmaUtil of the committed PMC passes (tools/make_traffic.py)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)          # >= 50 timed steps after >= 10 warm-up steps
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--batch-per-gpu", type=int, default=128)
    ap.add_argument("--model", default="deit_small_distilled_patch16_224")
    ap.add_argument("--wbits", type=int, default=2)
    ap.add_argument("--abits", type=int, default=2)
    ap.add_argument("--no-qkr", action="store_true")
    ap.add_argument("--cga", action="store_true", help="add the CGA mask/restore hooks (config C5)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-recipe-line", action="store_true",
                    help="skip the auxiliary measurement of the KD recipe step (fp32 teacher forward inside the step, train.py:906-910)")
    ap.add_argument("--no-c1-baseline", action="store_true", help="skip the second CPU line (BASELINE config 1, DeiT-T W4A4 B=32)")
    ap.add_argument("--cpu-batch", type=int, default=8)
    ap.add_argument("--cpu-steps", type=int, default=4)
    ap.add_argument("--no-roofline-events", action="store_true")
    ap.add_argument("--verbose", action="store_true")
    ap.add_argument("--with-teacher", action="store_true",
                    help="second line of SURVEY 8(d): add the fp32 teacher forward (no_grad, same architecture, random init) "
                         "that produces the KD soft targets to every step (train.py:906-910)")
    ap.add_argument("--no-graph", action="store_true",
                    help="launch every kernel from Python each step instead of replaying the captured hipGraph of the step")
    ap.add_argument("--graph", action="store_true",
                    help="with several ranks: capture the RCCL collectives inside the step's graph as well (default there: "
                         "captured compute in sub-graphs cut at the gradient buckets, each bucket's all-reduce issued eagerly "
                         "behind its sub-graph; the capture of collectives is validated with one rank only, tests/test_graph_gpu.py)")
    ap.add_argument("--split-graph", action="store_true",
                    help="with several ranks: round 4's form -- one graph for the whole backward, then every bucket all-reduce "
                         "(exposed), then the optimiser graph -- instead of the sub-graphs cut at the bucket boundaries")
    ap.add_argument("--teacher-gemm", default="f16x4", choices=["f32", "bf16x9", "bf16x6", "f16x4"],
                    help="with --with-teacher: fp32-MFMA GEMMs, or the weights pre-split into bf16 planes (9 / 6 plane products)")
    ap.add_argument("--stock-teacher", action="store_true",
                    help="with --with-teacher: run the teacher through stock PyTorch-ROCm (hipBLASLt) instead of the HIP kernels")
    ap.add_argument("--sync-statsq", action="store_true",
                    help="also all-reduce the StatsQ scale vectors each step and assert that it changes nothing (north_star's "
                         "'StatsQ statistics' collective: not in the reference, a no-op by construction)")
    ap.add_argument("--share-gpu", action="store_true",
                    help="development only: the N ranks of --gpus N all use device 0 and talk over gloo (device tensors through the "
                         "host) -- exercises the several-rank flow of this file and of the data-parallel step on a one-GPU box; the "
                         "number it prints is N ranks time-sharing one GPU, not a scaling point")
    ap.add_argument("--force-dp", action="store_true",
                    help="use the DataParallel wrapper (bucket hooks + RCCL all-reduce) even with one rank")
    return ap.parse_args()


def cpu_baseline(model, args):
    """Time the oracle (kind 'port': the CPU restatement pinned to the reference by tests/golden) on host cores."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import ofq_oracle as O
    try:
        cores = len(os.sched_getaffinity(0))      # cores this process may actually use (cgroup / affinity aware)
    except AttributeError:
        cores = os.cpu_count() or 1
    cores = max(1, min(cores, 64))
    torch.set_num_threads(cores)
    sd = {k: v.detach().cpu().clone() for k, v in model.state_dict().items()}
    leaves = {k: (v.requires_grad_(True) if v.dtype.is_floating_point and "clip_val" not in k and "signed" not in k
                  else v) for k, v in sd.items()}
    m0 = model
    cfg = dict(depth=len(m0.blocks), num_heads=m0.blocks[0].attn.num_heads, patch=16, wbits=args.wbits,
               abits=args.abits, qkr=not args.no_qkr)
    B = args.cpu_batch
    g = torch.Generator().manual_seed(42)
    img = torch.randn(B, 3, 224, 224, generator=g)
    tgt = torch.randint(0, 1000, (B,), generator=g)
    soft = torch.randn(B, 1000, generator=g)
    params = [v for v in leaves.values() if v.requires_grad]
    opt = torch.optim.AdamW(params, lr=5.47e-4, weight_decay=0.05)
    times = []
    for it in range(args.cpu_steps + 1):
        t0 = time.perf_counter()
        opt.zero_grad(set_to_none=True)
        c, d = O.deit_forward(img, leaves, cfg, training=True)
        loss = O.kd_loss_soft_and_hard(c, d, tgt, soft)
        loss.backward()
        opt.step()
        times.append(time.perf_counter() - t0)
    times = sorted(times[1:])
    med = times[len(times) // 2]
    c1 = None
    if not args.no_c1_baseline:
        c1 = cpu_baseline_c1(cores)
    return {"value": round(B / med, 3), "unit": "images/s", "cores": cores, "kind": "port", "also": c1,
            "sample": "oracle/ofq_oracle.py (eager torch-CPU restatement of the reference path), %s W%dA%d%s, batch %d, "
                      "median of %d steps after 1 warm-up, fwd+bwd+AdamW, %d threads"
                      % (args.model, args.wbits, args.abits, "" if args.no_qkr else " QKR", B, args.cpu_steps, cores)}


def cpu_baseline_c1(cores):
    """BASELINE.json configs[0] / BASELINE.md section 3: DeiT-T W4A4, plain attention, batch 32, the reference's own
    CPU-runnable case, timed on the same host cores through the oracle (fwd + bwd + AdamW, median of 3 after 1 warm-up)."""
    import ofq_oracle as O
    from ofq_amd import engine
    dev = torch.device("cuda", 0)
    m = engine.build_student("deit_tiny_distilled_patch16_224", 4, 4, qk_reparam=False).to(dev)
    g = torch.Generator().manual_seed(42)
    img = torch.randn(32, 3, 224, 224, generator=g)
    tgt = torch.randint(0, 1000, (32,), generator=g)
    soft = torch.randn(32, 1000, generator=g)
    engine.setup_alpha(m, img.to(dev))
    sd = {k: v.detach().cpu().clone() for k, v in m.state_dict().items()}
    del m
    leaves = {k: (v.requires_grad_(True) if v.dtype.is_floating_point and "clip_val" not in k and "signed" not in k
                  else v) for k, v in sd.items()}
    cfg = dict(depth=12, num_heads=3, patch=16, wbits=4, abits=4, qkr=False)
    opt = torch.optim.AdamW([v for v in leaves.values() if v.requires_grad], lr=5e-4, weight_decay=0.05)
    times = []
    for it in range(4):
        t0 = time.perf_counter()
        opt.zero_grad(set_to_none=True)
        c, d = O.deit_forward(img, leaves, cfg, training=True)
        O.kd_loss_soft_and_hard(c, d, tgt, soft).backward()
        opt.step()
        times.append(time.perf_counter() - t0)
    med = sorted(times[1:])[1]
    return {"value": round(32 / med, 3), "unit": "images/s", "cores": cores, "kind": "port",
            "sample": "BASELINE config 1: deit_tiny W4A4 plain attention, batch 32, oracle, median of 3 steps after 1 warm-up"}


class _c_stdout_to_stderr:
    """RCCL printf()s its NCCL_DEBUG=VERSION banner (exported on this pool) to the C stdout, which would land next to the
    one JSON line this program owes on stdout: point fd 1 at stderr while the communicator is created, flush libc's
    buffer, and put fd 1 back."""

    def __enter__(self):
        sys.stdout.flush()
        self.saved = os.dup(1)
        os.dup2(2, 1)
        return self

    def __exit__(self, *exc):
        try:
            import ctypes
            ctypes.CDLL(None).fflush(None)
        finally:
            os.dup2(self.saved, 1)
            os.close(self.saved)
        return False


def self_launch(args):
    """`python bench.py --gpus N` (N > 1) without a launcher: start N ranks of this file, one per GPU, as CHILD processes --
    before this process has touched the GPU (a process that has initialised HIP must never exec or be replaced) -- with the
    rendezvous variables torch.distributed.run would set, pass rank 0's JSON line through and exit with the worst code."""
    import socket
    import subprocess
    n = args.gpus
    have = torch.cuda.device_count()                 # counts devices without initialising the runtime
    if have < n and not (args.share_gpu and have >= 1):
        raise SystemExit("bench.py --gpus %d: only %d HIP device(s) visible" % (n, have))
    sock = socket.socket()
    sock.bind(("127.0.0.1", 0))
    port = sock.getsockname()[1]
    sock.close()
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), OFQ_BENCH_CHILD="1")
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL))
    # rank 0's stdout is read by a thread (its pipe must drain) while the parent watches every child: a rank that dies
    # would leave the others waiting in a collective for ever, so the first failure ends the rest (exact PIDs only)
    import threading
    chunks = []
    reader = threading.Thread(target=lambda: chunks.append(procs[0].stdout.read()), daemon=True)
    reader.start()
    failed = None
    while failed is None and any(p.poll() is None for p in procs):
        for r, p in enumerate(procs):
            rc = p.poll()
            if rc not in (None, 0):
                failed = (r, rc)
                break
        time.sleep(0.2)
    if failed is not None:
        for p in procs:
            if p.poll() is None:
                p.terminate()
        for p in procs:
            try:
                p.wait(timeout=20)
            except subprocess.TimeoutExpired:
                p.kill()
        print("bench.py: rank %d exited with code %d; the other ranks were stopped" % failed, file=sys.stderr)
    reader.join(timeout=10)
    sys.stdout.write(b"".join(chunks).decode())
    sys.stdout.flush()
    codes = [p.wait() for p in procs]
    raise SystemExit(max(abs(c) for c in codes) if failed is None else (abs(failed[1]) or 1))


def _child_line(extra_args, env_extra, what):
    """One more measurement next to the headline, by a CHILD process of this script after the headline's timed region (a second
    context on the GPU while the parent still holds its model: ~25 GB resident of 288).  Returns the child's numbers, or an
    `error` object carrying the tail of the child's stderr (an auxiliary line must not take the headline down, nor fail silently)."""
    import subprocess
    cmd = [sys.executable, os.path.abspath(__file__)] + extra_args + ["--steps", "20", "--warmup", "5", "--no-cpu-baseline",
                                                                      "--no-roofline-events", "--no-recipe-line"]
    err = b""
    try:
        r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300, cwd=ROOT, env=dict(os.environ, **env_extra))
        err = r.stderr
        d = json.loads(r.stdout.decode().strip().splitlines()[-1])
        return {"value": d["value"], "unit": d["unit"], "ms_per_step": d["ms_per_step"], "steps": d["steps"], "dtype": d["dtype"],
                "grad_planes": d["config"].get("grad_planes"), "what": what + " (" + d["config"]["workload"] + ")"}
    except Exception as e:                                           # noqa: BLE001
        return {"error": "%s: %s" % (type(e).__name__, e), "stderr_tail": err.decode("utf-8", "replace")[-600:]}


def recipe_step_line():
    """Auxiliary, outside the timed region: the same step with the reference recipe's fp32 teacher forward in it (train.py:906-910
    calls `teacher(input)` every step; the headline metric feeds synthetic teacher logits instead).  Reported next to
    the headline, never instead of it."""
    return _child_line(["--with-teacher"], {}, "the same step with the fp32 KD teacher's forward inside it: bench.py --with-teacher")


def exact_backward_line():
    """Auxiliary, outside the timed region: the same step with every backward code GEMM on its fp32-EXACT form (the gradient operand
    cut into three bf16 planes, OFQ_GRAD_PLANES=3) instead of the default two fp16 planes of the power-of-two-scaled tensor
    (fp32-grade on the tensor's scale, not element-wise fp32) -- the reference recipe is `amp: False`
    (configs/ours_imagenet_recipe.attn_q.yml:27), so the number on its own arithmetic is reported next to the headline."""
    return _child_line([], {"OFQ_GRAD_PLANES": "3"},
                       "the same step with exact-fp32-product backward GEMMs (3 bf16 planes): OFQ_GRAD_PLANES=3 python bench.py")


def main():
    args = parse()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        self_launch(args)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device: the OFQ MI355X path has no CPU fallback")
    if args.share_gpu:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    import torch.distributed as dist
    if world > 1 or args.force_dp:
        # RCCL writes its NCCL_DEBUG output (the pool exports NCCL_DEBUG=VERSION) to stdout by default; stdout carries
        # exactly one JSON line here, so send the library's chatter to stderr
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29517")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        with _c_stdout_to_stderr():
            if args.share_gpu:
                dist.init_process_group(backend="gloo")
            else:
                dist.init_process_group(backend="nccl", device_id=dev)
            dist.barrier()                                  # creates the communicator (and prints RCCL's banner) now
    if args.gpus != world and rank == 0 and world > 1:
        print("warning: --gpus %d but WORLD_SIZE %d" % (args.gpus, world), file=sys.stderr)

    from ofq_amd import engine, ops, parallel
    from ofq_amd.quantization.utils import KDLossSoftandHard

    def say(msg):
        if args.verbose and rank == 0:
            print("[bench %.1fs] %s" % (time.perf_counter() - T0, msg), file=sys.stderr, flush=True)
    T0 = time.perf_counter()
    B = args.batch_per_gpu
    model = engine.build_student(args.model, args.wbits, args.abits, qk_reparam=not args.no_qkr,
                                 qk_reparam_type=1 if args.cga else 0).to(dev)
    g = torch.Generator(device=dev).manual_seed(42 + rank)          # seed 42 + rank
    images = torch.randn(B, 3, 224, 224, device=dev, generator=g)
    target = torch.randint(0, 1000, (B,), device=dev, generator=g)
    soft = torch.randn(B, 1000, device=dev, generator=g)           # stands in for the fp32 teacher's logits
    say("model built")
    engine.setup_alpha(model, images)                               # creates every LSQ step (train.py:657)
    torch.cuda.synchronize()
    say("setup_alpha done")
    model.train()
    dp = (parallel.DataParallel(model, bucket_mb=24.0, force_sync=args.force_dp, sync_statsq=args.sync_statsq)
          if (world > 1 or args.force_dp) else None)
    opt = engine.make_optimizer(model, lr=5.47e-4, weight_decay=0.05)
    cga = engine.CGAHooks(model, args.wbits, 0.005, qk_reparam=not args.no_qkr) if args.cga else None
    loss_fn = KDLossSoftandHard()

    teacher = None
    if args.with_teacher:
        from ofq_amd.deit import create_model
        # train.py:428-437, :526-531: the teacher is created and moved to the device, never put in eval mode: it returns
        # ((cls, dist), attn) and KLLossSoft distils from the cls logits (quantization/utils.py:46-47)
        teacher = create_model(args.model, num_classes=1000).to(dev)
        for p_ in teacher.parameters():
            p_.requires_grad_(False)
        if not args.stock_teacher:                  # the same forward on the HIP kernels (ofq_amd/teacher.py)
            from ofq_amd.teacher import HipTeacher
            teacher = HipTeacher(teacher, gemm=args.teacher_gemm)

    def soft_targets():
        if teacher is None:
            return soft
        with torch.no_grad():
            t_out, _ = teacher(images)
        return t_out[0] if isinstance(t_out, tuple) else t_out

    def step():
        return engine.train_step(model, opt, images, target, soft_targets(), loss_fn, dp=dp, cga=cga)

    eager_step = step
    # one rank: graph replay of the whole step.  Several ranks: graph replay of the compute with the bucket all-reduces issued
    # eagerly between two graphs (RCCL collectives inside a hipGraph could only be exercised with one rank on the one-GPU
    # development boxes: --graph), so that the host issues a handful of launches per step instead of ~850 next to RCCL's proxy
    # threads; --no-graph is the eager path (collectives overlapped with backward, 20 ms of Python per step).
    use_graph = not args.no_graph
    # with the data-parallel wrapper (several ranks, or --force-dp): "split" = captured compute + eager collectives
    # (engine.GraphedTrainStep) unless --graph asks for the collectives inside the graph
    # ("segmented": graph A cut at the bucket boundaries, each bucket's all-reduce issued behind its sub-graph and overlapped
    # with the next one; --split-graph keeps round 4's form: all collectives after the whole backward graph)
    graph_mode = "full" if (dp is None or args.graph) else ("split" if args.split_graph else "segmented")
    n_warm = args.warmup
    if use_graph:
        # same step, device side replayed from a hipGraph: the first two calls run eagerly, the third captures
        gstep = engine.GraphedTrainStep(model, opt, loss_fn, dp=dp, cga=cga, warmup=2, alias_inputs=True, mode=graph_mode)
        n_warm = max(args.warmup, gstep.warmup + 1)          # the capture must not fall into the timed region

        def step():
            return gstep(images, target, soft_targets())      # (the teacher runs eagerly; its logits are copied into the
                                                              # graph's static soft-target buffer)

    for i in range(n_warm):
        step()
        if args.verbose:
            torch.cuda.synchronize()
            say("warmup step %d done" % i)
    # throughput pass: EXACTLY --steps steps, no instrumentation (a pair of HIP events around every matrix-core launch --
    # ~600 per step -- costs 2-4 ms per step on this stack and would be charged to `value`)
    ops.TIMERS = None
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss = step()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if ops.nt_sk_error(dev):
        raise SystemExit("bench.py: a stream-K hand-off of the dX GEMM timed out (cut tile stored without a partial): results invalid")
    if dp is not None and dp.sync_statsq:
        dp.check_statsq_pending()                     # --sync-statsq: the asserted no-op (outside the timed region: a host sync)
    loss_value = float(loss.detach())                 # (the static loss tensor of the graph: read before any further step)
    # roofline pass: the same --steps steps again, live, with HIP events around every matrix-core kernel launch (on the
    # stream the kernels are launched on); rank 0 reports the dominant class
    timer = None if args.no_roofline_events else {}
    if timer is not None:
        # An event pair times [record, launch, record] on the stream: if the device has caught up with the host, the host's own
        # time between the two records (argument marshalling of the launch: 10-20 us) is charged to the kernel -- on a slow host
        # that moved the int8 forward class from 3.0 to 4.8 ms per step and made it the "dominant" one.  So every instrumented
        # eager step is queued BEHIND two replays of the captured step (40 ms of device work for 0.2 ms of host time): its
        # kernels wait in the queue and the pairs measure the kernels.
        prefill = step if step is not eager_step else None
        for _ in range(args.steps):
            if prefill is not None:
                ops.TIMERS = None
                prefill()
                prefill()
            ops.TIMERS = timer
            eager_step()                                    # HIP events cannot sit inside a replayed graph
        torch.cuda.synchronize()
        ops.TIMERS = None
        if world > 1:
            dist.barrier()
    if world > 1:
        t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    say("timed region done: %.1f ms/step" % (1000.0 * elapsed / args.steps))
    ms_per_step = 1000.0 * elapsed / args.steps
    value = B * world * args.steps / elapsed

    out = None
    if rank == 0:
        roof = None
        if timer:
            sums = {k: v.summary() for k, v in timer.items()}
            name = max(sums, key=lambda k: sums[k]["total_ms"])          # dominant matrix-core kernel class by time
            sm = sums[name]
            peak = [v for k, v in PEAKS.items() if name.startswith(k)][0]
            achieved = sm["total_units"] / (sm["total_ms"] * 1e-3) / 1e12 if sm["total_ms"] > 0 else 0.0
            # HBM bytes per launch of this kernel class from the committed PMC passes -- only if they were taken on THESE
            # kernels (content hash of csrc/ + the header, ofq_amd/build.py); stale counters are not reported
            traffic, mfma_util, traffic_note = None, None, "no profiles/" + PROFILE_JSON
            try:
                from ofq_amd import build as _build
                with open(os.path.join(ROOT, "profiles", PROFILE_JSON)) as fh:
                    tj = json.load(fh)
                if tj.get("source_hash") == _build.source_hash():
                    traffic = tj.get(name.split(" ")[0], {}).get("traffic_bytes_per_launch")
                    mfma_util = tj.get(name.split(" ")[0], {}).get("mfma_util_pmc")
                    traffic_note = "profiles/%s (kernel sources %s)" % (PROFILE_JSON, tj.get("source_hash"))
                else:
                    traffic_note = ("profiles/%s was measured on kernel sources %s, this library is %s: not reported"
                                    % (PROFILE_JSON, tj.get("source_hash"), _build.source_hash()))
            except (OSError, ValueError):
                pass
            roof = {"kernel": name, "bound": "mfma", "achieved": round(achieved, 2), "peak": peak, "unit": "TFLOP/s",
                    "frac": round(achieved / peak, 4), "traffic": traffic,
                    # the split kernels issue two fp16 (round 5) or three bf16 MFMA FMAs per algorithmic FMA
                    "mfma_pipe_frac": round((2.0 if "2x v_mfma" in name else 3.0 if name.startswith("qgemm_bf16s") else 1.0) * achieved / peak, 4),
                    "mfma_util_pmc": mfma_util,
                    "note": "achieved = algorithmic 2*M*N*K of the launches / HIP-event time of the launches, over a second, "
                            "instrumented pass of the same --steps steps (the throughput pass carries no events: they cost "
                            "2-4 ms/step); the split kernels issue 2 fp16 (fp32-grade on the tensor's scale, ops.GRAD_PLANES = 2) or 3 bf16 (exact "
                            "fp32 product) MFMA FMAs per algorithmic FMA -- see the class name; traffic = "
                            "bytes per launch, 2*FETCH_SIZE + WRITE_SIZE of separate rocprofv3 --pmc passes: " + traffic_note,
                    "launches_per_step": sm["launches"] / args.steps, "avg_launch_ms": round(sm["avg_ms"], 4),
                    "avg_gflop_per_launch": round(sm["avg_units"] / 1e9, 3),
                    "ms_per_step": round(sm["total_ms"] / args.steps, 3),
                    "all_classes_ms_per_step": {k: round(v["total_ms"] / args.steps, 3) for k, v in sums.items()},
                    "all_classes_tflops": {k: round(v["total_units"] / (v["total_ms"] * 1e-3) / 1e12, 1) for k, v in sums.items()
                                           if v["total_ms"] > 0}}
        default_cfg = args.model == "deit_small_distilled_patch16_224" and args.wbits == 2 and args.abits == 2
        metric = ("images/sec QAT (DeiT-S W2A2, 224px synthetic)" if default_cfg      # BASELINE.json's metric string
                  else "images/sec QAT (%s W%dA%d, 224px synthetic)" % (args.model, args.wbits, args.abits))
        out = {"metric": metric, "value": round(value, 2), "unit": "images/s",
               "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_per_step, 3),
               "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
               # fp32 storage and accumulation everywhere; the precision trade of the backward GEMMs is part of the record
               "dtype": ("f32 (backward GEMM gradient operands as 2 fp16 planes of the 2^E-scaled tensor, fp32 accumulate)"
                         if ops.GRAD_PLANES == 2 else "f32 (backward GEMM gradient operands as 3 bf16 planes: exact fp32 products)"),
               "data": "synthetic",
               "config": {"workload": "%s W%dA%d%s%s QAT step (student fwd + KD loss + bwd + AdamW), %d img/GPU, "
                                      "%s, fp32, %s"
                                      % (args.model, args.wbits, args.abits, "" if args.no_qkr else " QKR",
                                         " + CGA hooks" if args.cga else "", B,
                                         "7x7 windows" if args.model.startswith("swin") else "198 tokens",
                                         ("fp32 teacher forward in the step (%s)" % ("stock PyTorch-ROCm" if args.stock_teacher else "HIP kernels, GEMMs " + args.teacher_gemm))
                                         if args.with_teacher else "teacher logits synthetic"),
                          "global_batch": B * world, "parallelism": "dp%d" % world, "loss": float(loss_value),
                          "grad_planes": ops.GRAD_PLANES,
                          "launch": ("eager (one ctypes launch per kernel)" if not use_graph else "hipGraph replay" if graph_mode == "full"
                                     else "hipGraph replay of the compute, bucket all-reduces eager between two graphs" if graph_mode == "split"
                                     else "hipGraph replay in %d sub-graphs cut at the gradient buckets, each bucket's all-reduce issued "
                                          "behind its sub-graph (overlapped with the next)" % len(gstep.segments)),
                          "rccl_ranks": (dist.get_world_size() if dist.is_initialized() else 1) if not args.share_gpu else 0,
                          **({"share_gpu": "%d ranks time-share ONE GPU over gloo (development run, not a scaling point)" % world}
                             if args.share_gpu else {})},
               "roofline": roof}
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(model, args)
        if (world == 1 and default_cfg and not args.with_teacher and not args.no_recipe_line and not args.no_cpu_baseline
                and use_graph and not args.cga and not args.force_dp):
            out["recipe_step"] = recipe_step_line()
            if ops.GRAD_PLANES == 2:
                out["exact_fp32_backward"] = exact_backward_line()
        print(json.dumps(out), flush=True)
    if dist.is_initialized():
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
