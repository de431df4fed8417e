#!/usr/bin/env python3
"""Which gradients keep DataParallel's bucket pack off torch._foreach_copy_'s fast route? (one rank over RCCL)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
os.environ.setdefault("RANK", "0"); os.environ.setdefault("WORLD_SIZE", "1")
import torch, torch.distributed as dist
from ofq_amd import engine, parallel
from ofq_amd.quantization.utils import KDLossSoftandHard
torch.cuda.set_device(0)
dist.init_process_group("nccl", device_id=torch.device("cuda", 0))
model = engine.build_student("deit_small_distilled_patch16_224", 2, 2, qk_reparam=True).cuda()
x = torch.randn(16, 3, 224, 224, device="cuda"); y = torch.randint(0, 1000, (16,), device="cuda"); s = torch.randn(16, 1000, device="cuda")
engine.setup_alpha(model, x); model.train()
dp = parallel.DataParallel(model, force_sync=True)
opt = engine.make_optimizer(model)
names = {id(p): n for n, p in model.named_parameters()}
orig = dp._launch
def probe(b):
    odd = [(names[id(p)], tuple(p.grad.shape), p.grad.stride(), p.grad.dtype) for p in b.params
           if p.grad is not None and not p.grad.is_contiguous()]
    print("bucket of %d params, %d non-contiguous grads: %s" % (len(b.params), len(odd), odd[:6]))
    return orig(b)
dp._launch = probe
for i in range(3):
    print("step", i)
    engine.train_step(model, opt, x, y, s, KDLossSoftandHard(), dp=dp)
torch.cuda.synchronize()
dist.destroy_process_group()
