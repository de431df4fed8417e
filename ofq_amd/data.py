"""On-device input pipeline (reference train.py:579-629 builds it from timm 0.5.4: FastCollateMixup
as the loader's collate, PrefetchLoader's normalisation and RandomErasing on the device).

What runs where, reference vs here:
  * JPEG decode, random-resized crop, flip, RandAugment (`aa: rand-m9-mstd0.5-inc1`): PIL operations in timm's CPU worker
    processes -- not part of this tier (no dataset on the box); the pipeline starts from the decoded uint8 batch
    [B][3][224][224], which is what fast_collate hands to the device in the reference too.
  * mixup / cutmix: timm mixes on the HOST inside the collate (numpy, uint8 space).  Here the same arithmetic runs on the
    device in the pass that normalises the batch.
  * normalisation + RandomErasing(mode='pixel'): on the device in both.
The random DECISIONS are host-side, exactly timm's draws in timm's order (numpy.random for mixup / cutmix, python `random`
for the erasing rectangles), so a seeded run reproduces the reference's augmentation parameters; only the erasing noise
comes from the device generator (as in timm, whose `torch.empty(...).normal_()` runs on the GPU as well).
One HIP launch per batch (csrc/input_pipeline.hip): 1-2 B read + 4 B written per element."""
import math
import random

import numpy as np
import torch

from . import ops

IMAGENET_DEFAULT_MEAN = (0.485, 0.456, 0.406)          # timm.data.constants
IMAGENET_DEFAULT_STD = (0.229, 0.224, 0.225)


class MixupParams:
    """timm.data.mixup.Mixup / FastCollateMixup parameter draws for mode='batch' (train.py:584-590)."""

    def __init__(self, mixup_alpha=0.8, cutmix_alpha=1.0, prob=1.0, switch_prob=0.5, label_smoothing=0.1, num_classes=1000,
                 correct_lam=True):
        self.mixup_alpha, self.cutmix_alpha = mixup_alpha, cutmix_alpha
        self.mix_prob, self.switch_prob = prob, switch_prob
        self.label_smoothing, self.num_classes, self.correct_lam = label_smoothing, num_classes, correct_lam
        self.mixup_enabled = True

    def params_per_batch(self):
        """Mixup._params_per_batch: (lam, use_cutmix); consumes numpy's global generator like timm."""
        lam, use_cutmix = 1.0, False
        if self.mixup_enabled and np.random.rand() < self.mix_prob:
            if self.mixup_alpha > 0.0 and self.cutmix_alpha > 0.0:
                use_cutmix = np.random.rand() < self.switch_prob
                lam_mix = (np.random.beta(self.cutmix_alpha, self.cutmix_alpha) if use_cutmix
                           else np.random.beta(self.mixup_alpha, self.mixup_alpha))
            elif self.mixup_alpha > 0.0:
                lam_mix = np.random.beta(self.mixup_alpha, self.mixup_alpha)
            elif self.cutmix_alpha > 0.0:
                use_cutmix = True
                lam_mix = np.random.beta(self.cutmix_alpha, self.cutmix_alpha)
            else:
                raise ValueError("One of mixup_alpha > 0., cutmix_alpha > 0. should be true.")
            lam = float(lam_mix)
        return lam, bool(use_cutmix)

    def draw(self, img_h, img_w):
        """(lam, use_cutmix, (yl, yh, xl, xh)) for one batch: _mix_batch_collate's draws incl. cutmix_bbox_and_lam / rand_bbox."""
        lam, use_cutmix = self.params_per_batch()
        box = (0, 0, 0, 0)
        if use_cutmix:
            ratio = np.sqrt(1 - lam)                                             # rand_bbox
            cut_h, cut_w = int(img_h * ratio), int(img_w * ratio)
            cy = np.random.randint(0, img_h)
            cx = np.random.randint(0, img_w)
            yl, yh = int(np.clip(cy - cut_h // 2, 0, img_h)), int(np.clip(cy + cut_h // 2, 0, img_h))
            xl, xh = int(np.clip(cx - cut_w // 2, 0, img_w)), int(np.clip(cx + cut_w // 2, 0, img_w))
            box = (yl, yh, xl, xh)
            if self.correct_lam:
                lam = 1.0 - (yh - yl) * (xh - xl) / float(img_h * img_w)
        return lam, use_cutmix, box

    def targets(self, target, lam):
        """mixup_target: label-smoothed one-hot of the sample and of its mirror partner, mixed with lam."""
        off = self.label_smoothing / self.num_classes
        on = 1.0 - self.label_smoothing + off
        t = target.long().view(-1, 1)
        y1 = torch.full((t.shape[0], self.num_classes), off, device=target.device).scatter_(1, t, on)
        y2 = torch.full((t.shape[0], self.num_classes), off, device=target.device).scatter_(1, t.flip(0), on)
        return y1 * lam + y2 * (1.0 - lam)


class RandomErasingParams:
    """timm.data.random_erasing.RandomErasing._erase's rectangle draws (python `random`), mode 'pixel', one rectangle."""

    def __init__(self, probability=0.25, min_area=0.02, max_area=1 / 3, min_aspect=0.3, max_aspect=None, min_count=1,
                 max_count=None):
        self.probability, self.min_area, self.max_area = probability, min_area, max_area
        max_aspect = max_aspect or 1 / min_aspect
        self.log_aspect_ratio = (math.log(min_aspect), math.log(max_aspect))
        self.min_count, self.max_count = min_count, max_count or min_count
        if self.max_count != 1:
            raise ValueError("recount > 1 is not used by the OFQ recipes (configs/*.yml: recount 1)")

    def draw(self, B, img_h, img_w):
        """int32 [B][4] {top, left, h, w}; h == 0: the sample is not erased."""
        rects = np.zeros((B, 4), dtype=np.int32)
        area = img_h * img_w
        for i in range(B):
            if random.random() > self.probability:
                continue
            count = self.min_count if self.min_count == self.max_count else random.randint(self.min_count, self.max_count)
            for _ in range(count):
                for _attempt in range(10):
                    target_area = random.uniform(self.min_area, self.max_area) * area / count
                    aspect_ratio = math.exp(random.uniform(*self.log_aspect_ratio))
                    h = int(round(math.sqrt(target_area * aspect_ratio)))
                    w = int(round(math.sqrt(target_area / aspect_ratio)))
                    if w < img_w and h < img_h:
                        top = random.randint(0, img_h - h)
                        left = random.randint(0, img_w - w)
                        rects[i] = (top, left, h, w)
                        break
        return rects


class DeviceInputPipeline:
    """uint8 batch [B][3][H][W] on the device -> (normalised fp32 batch, targets), one launch (see the module text).
    mixup=None / erasing=None switch the stages off; targets are soft ([B][classes]) when mixup is on, else the labels."""

    def __init__(self, mean=IMAGENET_DEFAULT_MEAN, std=IMAGENET_DEFAULT_STD, mixup=None, erasing=None):
        self.mean255 = np.array([x * 255 for x in mean], dtype=np.float32)     # PrefetchLoader: tensor([x * 255 for x in mean])
        self.std255 = np.array([x * 255 for x in std], dtype=np.float32)
        self.mixup, self.erasing = mixup, erasing

    def __call__(self, images_u8, target, noise=None):
        if not images_u8.is_cuda or images_u8.dtype != torch.uint8 or not images_u8.is_contiguous():
            raise RuntimeError("DeviceInputPipeline: contiguous uint8 [B][C][H][W] on a HIP device (no CPU fallback)")
        B, C, H, W = images_u8.shape
        lam, use_cutmix, box, use_mix = 1.0, False, (0, 0, 0, 0), False
        if self.mixup is not None:
            if B % 2:
                raise ValueError("Batch size should be even when using this")      # timm's assertion text
            lam, use_cutmix, box = self.mixup.draw(H, W)
            use_mix = lam != 1.0
        rects_dev = None
        if self.erasing is not None:
            rects = self.erasing.draw(B, H, W)
            if rects[:, 2].any():
                rects_dev = torch.from_numpy(rects).to(images_u8.device, non_blocking=True)
                if noise is None:
                    noise = torch.randn((B, C, H, W), dtype=torch.float32, device=images_u8.device)
        out = torch.empty((B, C, H, W), dtype=torch.float32, device=images_u8.device)
        ops._chk(ops.lib().ofq_input_pipeline_u8(images_u8.data_ptr(), out.data_ptr(), B, C, H, W, self.mean255.ctypes.data,
                                                 self.std255.ctypes.data, int(use_mix), int(use_cutmix), float(np.float32(lam)),
                                                 float(np.float32(1.0 - lam)), box[0], box[1], box[2], box[3],
                                                 ops._p(rects_dev), ops._p(noise) if rects_dev is not None else 0,
                                                 ops._stream()), "ofq_input_pipeline_u8")
        if self.mixup is not None:
            target = self.mixup.targets(target, lam)
        self.last = {"lam": lam, "use_cutmix": use_cutmix, "box": box, "rects": None if rects_dev is None else rects}
        return out, target
