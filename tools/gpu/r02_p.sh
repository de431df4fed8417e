#!/bin/bash
set -u
O=gpurun_out/r02p; mkdir -p $O
export TMPDIR=/tmp
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v amdgpu.ids | tail -3
timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29611 bench.py --gpus 1 --steps 5 --warmup 3 --no-cpu-baseline --no-roofline-events 2>/dev/null | tail -1 | cut -c1-300
timeout 300 python bench.py --gpus 2 --steps 5 --warmup 3 2>&1 | tail -2 | cut -c1-200
timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29612 bench.py --gpus 1 --force-dp --steps 5 --warmup 3 --no-cpu-baseline --no-roofline-events 2>/dev/null | tail -1 | cut -c1-300
timeout 300 python bench.py --cga --steps 10 --warmup 4 --no-cpu-baseline --no-roofline-events 2>/dev/null | tail -1 | cut -c1-200
timeout 300 python cga.py --model deit_tiny_distilled_patch16_224 --batch-size 8 --steps-per-epoch 3 --val-steps 1 --epochs 0 --freeze_for_n_epochs 1 --aq-enable --aq-mode lsq --aq-per-channel --aq_clip_learnable --aq-bitw 2 --wq-enable --wq-per-channel --wq-bitw 2 --wq-mode statsq --model_type deit --quantized --pretrained_initialized --qk_reparam --qk_reparam_type 1 --boundaryRange 0.005 --log-interval 1 2>&1 | grep -v amdgpu.ids | tail -5
