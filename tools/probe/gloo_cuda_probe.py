import os, torch, torch.distributed as dist
os.environ["MASTER_ADDR"]="127.0.0.1"; os.environ["MASTER_PORT"]="29577"
dist.init_process_group("gloo", rank=0, world_size=1)
t=torch.ones(4,device="cuda")
try:
    dist.all_reduce(t); dist.broadcast(t, src=0); w=dist.all_reduce(t, async_op=True); w.wait()
    print("gloo on CUDA tensors OK", t)
except Exception as e:
    print("gloo CUDA FAIL:", repr(e)[:300])
