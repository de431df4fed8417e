"""LearnableBias / LearnableBias4img — src/quantization/modules/qbias.py:5-23.  Inside the Q-modules the
offsets are fused into the LSQ kernel; the standalone forward (plain torch add) exists for API parity."""
import torch
import torch.nn as nn


class LearnableBias(nn.Module):
    def __init__(self, out_chn):
        super().__init__()
        self.bias = nn.Parameter(torch.zeros(out_chn), requires_grad=True)

    def forward(self, x):
        return x + self.bias.expand_as(x)


class LearnableBias4img(nn.Module):
    def __init__(self, out_chn):
        super().__init__()
        self.bias = nn.Parameter(torch.zeros(out_chn), requires_grad=True)

    def forward(self, x):
        return x + self.bias.reshape(x.shape[-1], x.shape[-2]).expand_as(x)
