#!/usr/bin/env python3
"""QAT training entry point (counterpart of the reference's train.py).  Example, 2-bit DeiT-S with QKR
(train_scripts/deit_s/w2a2_deit_s.sh), synthetic data, one MI355X:

  python train.py --model deit_small_distilled_patch16_224 --batch-size 128 --lr 5.47e-4 --weight-decay 0.05 \
      --aq-enable --aq-mode lsq --aq-per-channel --aq_clip_learnable --aq-bitw 2 --wq-enable --wq-per-channel \
      --wq-bitw 2 --wq-mode statsq --model_type deit --quantized --pretrained_initialized --use-kd \
      --teacher deit_small_distilled_patch16_224 --kd_hard_and_soft 1 --qk_reparam --qk_reparam_type 0

8 GPUs of one node:  python -m torch.distributed.run --nproc-per-node 8 --master-addr 127.0.0.1 train.py ...
"""
from ofq_amd.train_cli import main

if __name__ == "__main__":
    main(cga=False)
