"""train.py / cga.py entry points — MI355X counterpart of the reference's timm-derived scripts
(train.py:81-384 flags, :444-858 main, :860-995 train_one_epoch, :997-1010 setup_alpha, :1012-1083 validate;
cga.py adds --boundaryRange / --freeze_for_n_epochs and the mask/restore hooks, cga.py:369-370, :953-1013).

Same flag names and YAML-then-CLI precedence (train.py:369-384) for everything the fake-quant path consumes.
What is re-authored rather than mirrored: one process per GPU is started by torchrun (RANK/LOCAL_RANK/WORLD_SIZE)
or, like the reference, spawned from --world_size/--visible_gpu; gradients use ofq_amd.parallel.DataParallel
(flat buckets over RCCL); the data pipeline is a device-resident synthetic ImageNet-shaped loader (there is no
dataset on the box; timm's loader / augmentation stack is the "next" row of SURVEY.md §8f)."""
import argparse
import json
import math
import os
import random
import sys
import time

import numpy as np

import torch
import torch.distributed as dist


def build_parser(cga=False):
    cp = argparse.ArgumentParser(add_help=False)
    cp.add_argument('-c', '--config', default='', type=str, metavar='FILE')
    p = argparse.ArgumentParser(description='OFQ QAT on MI355X' + (' (CGA fine-tune)' if cga else ''))
    p.add_argument('data_dir', nargs='?', default='synthetic', help='ignored: batches are synthetic')
    p.add_argument('--dataset', default='synthetic')
    p.add_argument('--model', default='deit_small_distilled_patch16_224')
    p.add_argument('--num-classes', type=int, default=1000)
    p.add_argument('-b', '--batch-size', type=int, default=128)
    p.add_argument('--epochs', type=int, default=1)
    p.add_argument('--steps-per-epoch', type=int, default=20)
    p.add_argument('--val-steps', type=int, default=2)
    p.add_argument('--opt', default='adamw')
    p.add_argument('--lr', type=float, default=5e-4)
    p.add_argument('--min-lr', type=float, default=1e-5)
    p.add_argument('--warmup-lr', type=float, default=1e-6)
    p.add_argument('--warmup-epochs', type=int, default=0)
    p.add_argument('--weight-decay', type=float, default=0.05)
    p.add_argument('--sched', default='cosine')
    p.add_argument('--seed', type=int, default=42)
    p.add_argument('--log-interval', type=int, default=50)
    p.add_argument('--cooldown-epochs', type=int, default=0)
    p.add_argument('--smoothing', type=float, default=0.1)
    p.add_argument('--output', default='')
    p.add_argument('--resume', default='')
    p.add_argument('--no-resume-opt', action='store_true')
    p.add_argument('--initial-checkpoint', default='', help='student weights loaded after the surgery, strict=False (train.py:515-516)')
    # on-device input pipeline (train.py:199-220, :579-629; ofq_amd/data.py)
    p.add_argument('--mixup', type=float, default=0.0)
    p.add_argument('--cutmix', type=float, default=0.0)
    p.add_argument('--mixup-prob', type=float, default=1.0)
    p.add_argument('--mixup-switch-prob', type=float, default=0.5)
    p.add_argument('--mixup-mode', default='batch')
    p.add_argument('--mixup-off-epoch', type=int, default=0)
    p.add_argument('--reprob', type=float, default=0.0)
    p.add_argument('--remode', default='pixel')
    p.add_argument('--recount', type=int, default=1)
    p.add_argument('--aa', default=None, help='RandAugment runs in timm CPU workers on PIL images: not part of this path')
    p.add_argument('--no-graph', action='store_true',
                   help='launch every kernel from Python instead of replaying the captured hipGraph of the step')
    p.add_argument('--sync-statsq', action='store_true',
                   help='all-reduce the StatsQ scale vectors behind the last gradient bucket of every step and assert that it '
                        'is a no-op (they are functions of replica-identical weights; the reference has no such collective)')
    p.add_argument('--graph', action='store_true',
                   help='capture the step in a hipGraph also when several ranks train together (the captured RCCL all-reduce '
                        'has only been exercised with one rank: without this flag multi-rank runs launch eagerly, like bench.py)')
    # quantisation flags (train.py:297-366)
    p.add_argument('--quantized', action='store_true')
    p.add_argument('--wq-enable', action='store_true')
    p.add_argument('--wq-mode', default='statsq')
    p.add_argument('--wq-bitw', type=int, default=2)
    p.add_argument('--wq-per-channel', action='store_true')
    p.add_argument('--wq-asym', action='store_true')
    p.add_argument('--wq_clip_learnable', action='store_true')
    p.add_argument('--aq-enable', action='store_true')
    p.add_argument('--aq-mode', default='lsq')
    p.add_argument('--aq-bitw', type=int, default=2)
    p.add_argument('--aq-per-channel', action='store_true')
    p.add_argument('--aq_clip_learnable', action='store_true')
    p.add_argument('--act-layer', default='gelu')
    p.add_argument('--model_type', default='deit')
    p.add_argument('--pretrained', action='store_true')
    p.add_argument('--pretrained_initialized', action='store_true')
    p.add_argument('--qk_reparam', action='store_true')
    p.add_argument('--qk_reparam_type', type=int, default=0)
    p.add_argument('--qmodules', nargs='*', default=None)
    # distillation (train.py:428-442, :906-910)
    p.add_argument('--use-kd', action='store_true')
    p.add_argument('--teacher', default='deit_small_distilled_patch16_224')
    p.add_argument('--teacher_pretrained', action='store_true')
    p.add_argument('--teacher-checkpoint', default='', help='teacher weights, strict (train.py:331, :440-441)')
    p.add_argument('--teacher-random-init', action='store_true',
                   help='explicitly accept a randomly initialised teacher (synthetic-data smoke runs only)')
    p.add_argument('--kd_hard_and_soft', type=int, default=0, help='0: soft label only, 1: hard + soft label (train.py:360)')
    # process layout (train.py:1085-1096)
    p.add_argument('--world_size', default='1')
    p.add_argument('--visible_gpu', default='')
    p.add_argument('--tcp_port', default='36969')
    # CGA (cga.py:369-370)
    p.add_argument('--boundaryRange', type=float, default=0.005)
    p.add_argument('--freeze_for_n_epochs', type=int, default=30 if cga else 0)
    return cp, p


def parse_args(argv, cga):
    cp, p = build_parser(cga)
    cfg_args, remaining = cp.parse_known_args(argv)
    if cfg_args.config:
        import yaml
        with open(cfg_args.config) as f:
            cfg = yaml.safe_load(f)
        known = {a.dest for a in p._actions}
        p.set_defaults(**{k.replace('-', '_'): v for k, v in cfg.items() if k.replace('-', '_') in known})
    return p.parse_args(remaining)


class SyntheticLoader:
    """Device-resident ImageNet-shaped batches (randn images, random labels), sharded by rank through the seed.
    With a `pipeline` (ofq_amd.data.DeviceInputPipeline) the pool holds decoded uint8 batches -- what timm's fast_collate
    hands to the device -- and every batch goes through mixup / cutmix, normalisation and random erasing on the device."""

    def __init__(self, steps, batch, num_classes, device, seed, pipeline=None):
        self.steps, self.batch, self.nc, self.device, self.pipeline = steps, batch, num_classes, device, pipeline
        g = torch.Generator(device=device).manual_seed(seed)
        if pipeline is None:
            self.pool = [(torch.randn(batch, 3, 224, 224, device=device, generator=g),
                          torch.randint(0, num_classes, (batch,), device=device, generator=g)) for _ in range(2)]
        else:
            self.pool = [(torch.randint(0, 256, (batch, 3, 224, 224), device=device, generator=g, dtype=torch.uint8),
                          torch.randint(0, num_classes, (batch,), device=device, generator=g)) for _ in range(2)]

    def __len__(self):
        return self.steps

    def __iter__(self):
        for i in range(self.steps):
            x, y = self.pool[i % len(self.pool)]
            yield (x, y) if self.pipeline is None else self.pipeline(x, y)


def cosine_lr(epoch, epochs, base, min_lr, warmup_epochs, warmup_lr):
    """timm 0.5.4 CosineLRScheduler as train.py:731-739, :840 drives it (t_in_epochs, warmup_prefix=False, one cycle):
    the rate is set once per EPOCH; linear warm-up from warmup_lr for warmup_epochs, then the cosine on the UNSHIFTED
    epoch index, and min_lr from `epochs` on (the cool-down epochs)."""
    if epoch < warmup_epochs:
        return warmup_lr + epoch * (base - warmup_lr) / warmup_epochs
    if epoch >= epochs:
        return min_lr
    return min_lr + 0.5 * (base - min_lr) * (1 + math.cos(math.pi * epoch / epochs))


def load_checkpoint(model, path, strict=True, map_location="cpu"):
    """timm.models.helpers.load_checkpoint (train.py:441, :516): accepts a bare state dict or a training checkpoint
    ('state_dict' / 'model'), strips DDP's 'module.' prefix."""
    ck = torch.load(path, map_location=map_location, weights_only=False)
    sd = ck
    if isinstance(ck, dict):
        for key in ("state_dict", "model"):
            if key in ck and isinstance(ck[key], dict):
                sd = ck[key]
                break
    sd = {(k[7:] if k.startswith("module.") else k): v for k, v in sd.items()}
    return model.load_state_dict(sd, strict=strict)


def check_supported(args):
    """Flags the reference accepts that this path does not implement must not be dropped silently."""
    bad = []
    if args.opt.lower() != "adamw":
        bad.append("--opt %s (only adamw: train_scripts/*, configs/*.yml)" % args.opt)
    if args.sched != "cosine":
        bad.append("--sched %s (only cosine)" % args.sched)
    if (args.mixup > 0 or args.cutmix > 0) and args.mixup_mode != "batch":
        bad.append("--mixup-mode %s (only 'batch', the recipes' setting)" % args.mixup_mode)
    if args.reprob > 0 and (args.remode != "pixel" or args.recount != 1):
        bad.append("--remode %s --recount %d (only pixel / 1, the recipes' setting)" % (args.remode, args.recount))
    if (args.mixup > 0 or args.cutmix > 0) and args.batch_size % 2:
        bad.append("mixup / cutmix need an even batch size (timm: 'Batch size should be even when using this')")
    if args.use_kd and args.kd_hard_and_soft not in (0, 1):
        bad.append("--kd_hard_and_soft %d (0: soft only, 1: hard + soft)" % args.kd_hard_and_soft)
    if args.use_kd and not args.teacher_checkpoint and not args.teacher_random_init:
        bad.append("--use-kd without --teacher-checkpoint: the teacher would distil noise (pretrained weights cannot be "
                   "downloaded here); pass --teacher-random-init to accept that for a synthetic smoke run")
    if bad:
        raise SystemExit("ofq_amd train: unsupported or inconsistent options:\n  " + "\n  ".join(bad))


def log(rank, msg):
    if rank == 0:
        print(msg, flush=True)


def main_worker(local_rank, args, cga, spawned):
    from . import engine, ops, parallel
    from .deit import create_model
    from .quantization.utils import KDLossSoftandHard, KLLossSoft
    check_supported(args)
    if spawned:
        world = int(args.world_size)
        rank = local_rank
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", str(args.tcp_port))
    else:
        world = int(os.environ.get("WORLD_SIZE", "1"))
        rank = int(os.environ.get("RANK", "0"))
        local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("the OFQ MI355X path needs a HIP device (no CPU fallback)")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)          # train.py:474
    torch.manual_seed(args.seed)                                                             # train.py:501
    model = create_model(args.model, num_classes=args.num_classes, pretrained=args.pretrained)   # raises: no download
    if args.quantized:
        if args.qmodules is None:
            if args.model_type == 'swin':
                args.qmodules = engine.default_qmodules_swin([len(st) for st in model.features[1::2]])
            else:
                args.qmodules = engine.default_qmodules(len(model.blocks))
        model = engine.get_qat_model(model, args)                                           # train.py:523
    model.to(dev)
    mixup_active = args.mixup > 0 or args.cutmix > 0                                        # train.py:582
    pipeline = mixup = None
    if mixup_active or args.reprob > 0:
        from . import data
        np.random.seed(args.seed + rank)                                                    # timm.utils.random_seed
        random.seed(args.seed + rank)
        if mixup_active:
            mixup = data.MixupParams(mixup_alpha=args.mixup, cutmix_alpha=args.cutmix, prob=args.mixup_prob,
                                     switch_prob=args.mixup_switch_prob, label_smoothing=args.smoothing,
                                     num_classes=args.num_classes)
        erasing = data.RandomErasingParams(probability=args.reprob) if args.reprob > 0 else None
        pipeline = data.DeviceInputPipeline(mixup=mixup, erasing=erasing)
    loader = SyntheticLoader(args.steps_per_epoch, args.batch_size, args.num_classes, dev, args.seed + rank, pipeline)
    val_loader = SyntheticLoader(args.val_steps, args.batch_size, args.num_classes, dev, args.seed + 1000 + rank)
    if args.quantized:
        first = loader.pool[0][0]
        if pipeline is not None:                                                            # a normalised batch without augmentation
            first = data.DeviceInputPipeline()(first, loader.pool[0][1])[0]
        engine.setup_alpha(model, first)                                                    # train.py:656-657
    if args.initial_checkpoint:                                                              # train.py:515-516
        # (the lazily created LSQ steps exist by now, so a quantised checkpoint's `s` vectors have somewhere to go)
        load_checkpoint(model, args.initial_checkpoint, strict=False, map_location=dev)
    teacher = None
    if args.use_kd:                                                                          # train.py:428-442, :526-531
        teacher = create_model(args.teacher, num_classes=args.num_classes, pretrained=args.teacher_pretrained).to(dev)
        if args.teacher_checkpoint:
            load_checkpoint(teacher, args.teacher_checkpoint, strict=True, map_location=dev)
        else:
            log(rank, "WARNING: --teacher-random-init: the teacher has random weights, the soft targets are noise")
        for p_ in teacher.parameters():          # the reference leaves them trainable and back-propagates into the teacher
            p_.requires_grad_(False)             # for nothing (no optimizer holds them); the student's gradients are the same
    log(rank, "model: %s, %.2f M parameters" % (args.model, sum(p.numel() for p in model.parameters()) / 1e6))
    optimizer = engine.make_optimizer(model, lr=args.lr, weight_decay=args.weight_decay)    # train.py:662
    start_epoch = 0
    if args.resume:                                                                         # train.py:691-706
        ck = torch.load(args.resume, map_location=dev, weights_only=False)
        model.load_state_dict({(k[7:] if k.startswith("module.") else k): v for k, v in ck["state_dict"].items()})
        if "optimizer" in ck and not args.no_resume_opt:
            optimizer.load_state_dict(ck["optimizer"])
        start_epoch = ck.get("epoch", -1) + 1
    dp = parallel.DataParallel(model, sync_statsq=args.sync_statsq) if world > 1 else None  # train.py:727
    kd_both = KDLossSoftandHard()
    kd_soft = KLLossSoft()
    # train.py:764-769: with mixup the smoothing is in the soft targets (SoftTargetCrossEntropy), else label smoothing
    hard = torch.nn.CrossEntropyLoss(label_smoothing=0.0 if mixup_active else args.smoothing)

    def loss_fn(out, target, soft):                                                          # train.py:896-913
        if teacher is None:
            return hard(out[0] if isinstance(out, tuple) else out, target)
        if args.kd_hard_and_soft == 0:
            return kd_soft(out, soft)
        return kd_both(out, target, soft)
    qkr = bool(args.qk_reparam)
    hooks = engine.CGAHooks(model, args.wq_bitw, args.boundaryRange, qk_reparam=qkr, model_type=args.model_type) if cga else None
    first, last = (args.epochs, args.epochs + args.freeze_for_n_epochs) if cga else (start_epoch, args.epochs + args.cooldown_epochs)
    graphed = None
    multi_rank = dp is not None and getattr(dp, "world", 1) > 1
    if not args.no_graph and hasattr(optimizer, "advance_for_replay"):
        # several ranks: captured compute in sub-graphs cut at the gradient buckets, each bucket's all-reduce issued eagerly behind
        # its sub-graph and overlapped with the next (--graph: collectives captured too)
        graphed = engine.GraphedTrainStep(model, optimizer, loss_fn, dp=dp, cga=hooks,
                                          mode="segmented" if (multi_rank and not args.graph) else "full")
    no_soft = torch.zeros(args.batch_size, args.num_classes, device=dev)
    for epoch in range(first, last):                                                        # cga.py:760 / train.py:816
        model.train()
        t_epoch = time.time()
        end = time.time()
        last_logged = -1
        # cga.py fine-tunes at min-lr; train.py: one rate per epoch (timm's scheduler steps per epoch, train.py:840)
        lr = args.min_lr if cga else cosine_lr(epoch, args.epochs, args.lr, args.min_lr, args.warmup_epochs, args.warmup_lr)
        for gparam in optimizer.param_groups:
            gparam["lr"] = lr
        if mixup is not None and args.mixup_off_epoch and epoch >= args.mixup_off_epoch:
            mixup.mixup_enabled = False                                                     # train.py:865-869
        for bi, (x, y) in enumerate(loader):
            if teacher is not None:
                with torch.no_grad():
                    soft, _ = teacher(x)          # train mode, like the reference: ((cls, dist), attn); KLLossSoft takes [0]
                    soft = soft[0] if isinstance(soft, tuple) else soft
            else:
                soft = no_soft
            if graphed is not None:
                loss = graphed(x, y, soft)
            else:
                loss = engine.train_step(model, optimizer, x, y, soft, loss_fn, dp=dp, cga=hooks)
            if bi % args.log_interval == 0 or bi == len(loader) - 1:
                torch.cuda.synchronize()                                                    # train.py:944
                if dp is not None and dp.sync_statsq:
                    dp.check_statsq_pending()     # --sync-statsq: raises when the replicas' StatsQ scales have drifted apart
                if ops.nt_sk_error(x.device):
                    raise RuntimeError("ofq_amd: a stream-K hand-off of the dX GEMM timed out; the gradients of this run are invalid")
                bt = time.time() - end
                lv = loss.detach().clone()
                if world > 1:
                    lv = parallel.reduce_tensor(lv, world)                                  # train.py:952
                n = bi - last_logged
                last_logged = bi
                log(rank, "Train: %d [%4d/%d]  Loss: %9.6f  Time: %.3fs, %7.2f/s  LR: %.3e"
                    % (epoch, bi, len(loader), float(lv), bt / n, x.size(0) * world * n / max(bt, 1e-9), lr))
                end = time.time()
        metrics = validate(model, val_loader, world, rank)                                  # train.py:828
        log(rank, "epoch %d done in %.1fs  val top1 %.2f  val loss %.4f" % (epoch, time.time() - t_epoch,
                                                                           metrics["top1"], metrics["loss"]))
        if args.output and rank == 0:                                                       # train.py:850 (timm CheckpointSaver)
            os.makedirs(args.output, exist_ok=True)
            torch.save({"epoch": epoch, "arch": args.model, "state_dict": model.state_dict(),
                        "optimizer": optimizer.state_dict(), "version": 2, "args": vars(args), "metric": metrics["top1"]},
                       os.path.join(args.output, "last.pth.tar"))
    if world > 1:
        dist.destroy_process_group()
    return {"model": model, "optimizer": optimizer, "metrics": metrics if last > first else None}


@torch.no_grad()
def validate(model, loader, world, rank):
    """train.py:1012-1083: eval-mode forward (averaged cls/dist logits), top-1 and CE loss, reduced over ranks."""
    model.eval()
    correct = torch.zeros((), device=next(model.parameters()).device)
    total = 0
    loss_sum = torch.zeros_like(correct)
    for x, y in loader:
        out, _ = model(x)
        loss_sum += torch.nn.functional.cross_entropy(out, y)
        correct += (out.argmax(1) == y).sum()
        total += y.numel()
    stats = torch.stack([correct, loss_sum])
    if world > 1:
        dist.all_reduce(stats)                                                              # train.py:1048-1050
        total *= world
    return {"top1": 100.0 * float(stats[0]) / max(total, 1), "loss": float(stats[1]) / max(len(loader) * world, 1)}


def main(argv=None, cga=False):
    args = parse_args(sys.argv[1:] if argv is None else argv, cga)
    if "WORLD_SIZE" in os.environ or int(args.world_size) <= 1:
        return main_worker(0, args, cga, spawned=False)
    else:
        if args.visible_gpu:
            os.environ["CUDA_VISIBLE_DEVICES"] = args.visible_gpu                           # train.py:1087
        import torch.multiprocessing as mp
        mp.spawn(main_worker, nprocs=int(args.world_size), args=(args, cga, True))          # train.py:1093-1096
