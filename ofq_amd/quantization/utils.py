"""Distillation losses used by the shipped recipes — src/quantization/utils.py:44-77 (KLLossSoft,
KDLossSoftandHard, `--kd_hard_and_soft 1`).  B x 1000 logits: negligible work, stock torch ops."""
import torch
import torch.nn as nn
import torch.nn.functional as F


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
            return self.KLSoft(dist_output, soft_target) + self.Hard(cls_output, hard_target)
        return self.KLSoft(output, soft_target) + self.Hard(output, hard_target)
