"""Distillation losses used by the shipped recipes — src/quantization/utils.py:44-77 (KLLossSoft,
KDLossSoftandHard, `--kd_hard_and_soft 1`).  KDLossSoftandHard on device logits runs as one HIP kernel pair
(functional.KDLossFn: value + both gradients, instead of ~20 ATen launches); everything else uses stock torch ops."""
import torch
import torch.nn as nn
import torch.nn.functional as F

try:                                   # the HIP path (device tensors); CPU tensors keep the stock ops below
    from .. import functional as F_ofq
except Exception:  # noqa: BLE001
    F_ofq = None


class KLLossSoft(torch.nn.modules.loss._Loss):
    def forward(self, output, target, T=1.0):
        output = output[0] if isinstance(output, tuple) else output
        target = target[0] if isinstance(target, tuple) else target
        output, target = output / T, target / T
        loss = -torch.sum(F.softmax(target, dim=1) * F.log_softmax(output, dim=1), dim=1)
        if self.reduction == "mean":
            return loss.mean()
        if self.reduction == "sum":
            return loss.sum()
        return loss


class KDLossSoftandHard(nn.Module):
    def __init__(self):
        super().__init__()
        self.KLSoft = KLLossSoft()
        self.Hard = nn.CrossEntropyLoss()

    def forward(self, output, hard_target, soft_target):
        if isinstance(output, tuple):
            cls_output, dist_output = output[0], output[1]
            soft = soft_target[0] if isinstance(soft_target, tuple) else soft_target
            if F_ofq is not None and F_ofq.kd_loss_fusable(cls_output, dist_output, soft, hard_target):
                return F_ofq.KDLossFn.apply(cls_output, dist_output, soft, hard_target)        # same value, two launches
            return self.KLSoft(dist_output, soft_target) + self.Hard(cls_output, hard_target)
        return self.KLSoft(output, soft_target) + self.Hard(output, hard_target)
