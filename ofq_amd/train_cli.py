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
import sys
import time

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
    p.add_argument('--output', default='')
    p.add_argument('--resume', default='')
    p.add_argument('--mixup', type=float, default=0.0)
    p.add_argument('--cutmix', type=float, default=0.0)
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
    p.add_argument('--kd_hard_and_soft', type=int, default=1)
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
    """Device-resident ImageNet-shaped batches (randn images, random labels), sharded by rank through the seed."""

    def __init__(self, steps, batch, num_classes, device, seed):
        self.steps, self.batch, self.nc, self.device = steps, batch, num_classes, device
        g = torch.Generator(device=device).manual_seed(seed)
        self.pool = [(torch.randn(batch, 3, 224, 224, device=device, generator=g),
                      torch.randint(0, num_classes, (batch,), device=device, generator=g)) for _ in range(2)]

    def __len__(self):
        return self.steps

    def __iter__(self):
        for i in range(self.steps):
            yield self.pool[i % len(self.pool)]


def cosine_lr(step, total, base, min_lr, warmup_steps, warmup_lr):
    if step < warmup_steps:
        return warmup_lr + (base - warmup_lr) * step / max(1, warmup_steps)
    t = (step - warmup_steps) / max(1, total - warmup_steps)
    return min_lr + 0.5 * (base - min_lr) * (1 + math.cos(math.pi * min(t, 1.0)))


def log(rank, msg):
    if rank == 0:
        print(msg, flush=True)


def main_worker(local_rank, args, cga, spawned):
    from . import engine, parallel
    from .deit import create_model
    from .quantization.utils import KDLossSoftandHard
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
    model = create_model(args.model, num_classes=args.num_classes)
    if args.quantized:
        if args.qmodules is None:
            if args.model_type == 'swin':
                args.qmodules = engine.default_qmodules_swin([len(st) for st in model.features[1::2]])
            else:
                args.qmodules = engine.default_qmodules(len(model.blocks))
        model = engine.get_qat_model(model, args)                                           # train.py:523
    model.to(dev)
    teacher = None
    if args.use_kd:
        teacher = create_model(args.teacher, num_classes=args.num_classes).to(dev)          # train.py:428
    loader = SyntheticLoader(args.steps_per_epoch, args.batch_size, args.num_classes, dev, args.seed + rank)
    val_loader = SyntheticLoader(args.val_steps, args.batch_size, args.num_classes, dev, args.seed + 1000 + rank)
    engine.setup_alpha(model, loader.pool[0][0])                                            # train.py:656-657
    log(rank, "model: %s, %.2f M parameters" % (args.model, sum(p.numel() for p in model.parameters()) / 1e6))
    optimizer = engine.make_optimizer(model, lr=args.lr, weight_decay=args.weight_decay)    # train.py:662
    start_epoch = 0
    if args.resume:                                                                         # train.py:691-706
        ck = torch.load(args.resume, map_location=dev)
        model.load_state_dict(ck["state_dict"])
        if "optimizer" in ck:
            optimizer.load_state_dict(ck["optimizer"])
        start_epoch = ck.get("epoch", -1) + 1
    dp = parallel.DataParallel(model) if world > 1 else None                                # train.py:727
    loss_fn = KDLossSoftandHard()
    qkr = bool(args.qk_reparam)
    hooks = engine.CGAHooks(model, args.wq_bitw, args.boundaryRange, qk_reparam=qkr, model_type=args.model_type) if cga else None
    first, last = (args.epochs, args.epochs + args.freeze_for_n_epochs) if cga else (start_epoch, args.epochs)
    total_steps = max(1, args.epochs * len(loader))
    for epoch in range(first, last):                                                        # cga.py:760 / train.py:816
        model.train()
        t_epoch = time.time()
        end = time.time()
        last_logged = -1
        for bi, (x, y) in enumerate(loader):
            step = epoch * len(loader) + bi
            lr = args.min_lr if cga else cosine_lr(step, total_steps, args.lr, args.min_lr,
                                                   args.warmup_epochs * len(loader), args.warmup_lr)
            for gparam in optimizer.param_groups:
                gparam["lr"] = lr
            if teacher is not None:
                with torch.no_grad():
                    soft, _ = teacher(x)
                    soft = soft[0] if isinstance(soft, tuple) else soft
            else:
                soft = torch.zeros(x.shape[0], args.num_classes, device=dev)
            loss = engine.train_step(model, optimizer, x, y, soft, loss_fn, dp=dp, cga=hooks)
            if bi % args.log_interval == 0 or bi == len(loader) - 1:
                torch.cuda.synchronize()                                                    # train.py:944
                bt = time.time() - end
                lv = loss.detach()
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
        if args.output and rank == 0:                                                       # train.py:850
            os.makedirs(args.output, exist_ok=True)
            torch.save({"epoch": epoch, "state_dict": model.state_dict(), "optimizer": optimizer.state_dict(),
                        "args": vars(args)}, os.path.join(args.output, "last.pth.tar"))
    if world > 1:
        dist.destroy_process_group()


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
        main_worker(0, args, cga, spawned=False)
    else:
        if args.visible_gpu:
            os.environ["CUDA_VISIBLE_DEVICES"] = args.visible_gpu                           # train.py:1087
        import torch.multiprocessing as mp
        mp.spawn(main_worker, nprocs=int(args.world_size), args=(args, cga, True))          # train.py:1093-1096
