"""fp32 Swin-T skeleton — host-side counterpart of src/swin.py (PatchMerging :26, ShiftedWindowAttention :176,
SwinTransformerBlock :255, SwinTransformer :330, swin_t :512) with the same module tree / state-dict names
(`features.0.0` = 4x4 patch conv, `features.{1,3,5,7}.{i}.{norm1,attn,norm2,mlp}`, `features.{2,4,6}.{reduction,norm}`),
so the reference's `qmodules` name lists and checkpoints carry over.  torchvision is not a dependency: `MLP` and
`Permute` are the few lines needed from it.  Blocks pass `(features, attn_info)` tuples like the reference."""
import os
from functools import partial
from typing import List

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import functional as F_ofq
from . import ops


class MLP(nn.Sequential):
    """torchvision.ops.misc.MLP layout: Linear, act, Dropout, Linear, Dropout  (indices 0..4)."""

    def __init__(self, in_channels, hidden_channels: List[int], activation_layer=nn.GELU, dropout=0.0):
        layers, d = [], in_channels
        for h in hidden_channels[:-1]:
            layers += [nn.Linear(d, h), activation_layer(), nn.Dropout(dropout)]
            d = h
        layers += [nn.Linear(d, hidden_channels[-1]), nn.Dropout(dropout)]
        super().__init__(*layers)


class Permute(nn.Module):
    def __init__(self, dims):
        super().__init__()
        self.dims = dims

    def forward(self, x):
        return torch.permute(x, self.dims)


_MERGE_PERM = {}
_SWIN_WQK = os.environ.get("OFQ_NO_SWIN_WQK") is None                # A/B switch: W_qk of a stage's blocks in one batched launch
_SWIN_FUSE = os.environ.get("OFQ_NO_SWIN_MLP_FUSE") is None      # A/B switch of round 6's LayerNorm + quantiser fusions in Swin


class PatchMerging(nn.Module):
    def __init__(self, dim, norm_layer=nn.LayerNorm):
        super().__init__()
        self.dim = dim
        self.reduction = nn.Linear(4 * dim, 2 * dim, bias=False)
        self.norm = norm_layer(4 * dim)

    @staticmethod
    def _gather4(fx):
        return torch.cat([fx[..., 0::2, 0::2, :], fx[..., 1::2, 0::2, :], fx[..., 0::2, 1::2, :], fx[..., 1::2, 1::2, :]], -1)

    def forward(self, x):
        fx, info = x
        H, W, C = fx.shape[-3:]
        if H % 2 == 0 and W % 2 == 0 and fx.is_cuda and fx.dim() == 4 and fx.dtype == torch.float32 and C % 4 == 0:
            # even maps (every stage of the 224-px recipes): the 2x2 neighbourhood gather is a permutation of the tokens --
            # one row-gather kernel each way instead of pad + four strided slices + cat (and four slice_backward + three
            # adds on the way back); the index is built by pushing the token numbers through the very same slicing code
            key = (H, W, str(fx.device))
            hit = _MERGE_PERM.get(key)
            if hit is None:
                grid = torch.arange(H * W, device=fx.device, dtype=torch.float32).view(1, H, W, 1)
                idx = self._gather4(grid).reshape(-1).long()
                inv = torch.empty_like(idx)
                inv[idx] = torch.arange(idx.numel(), device=fx.device)
                hit = _MERGE_PERM[key] = (idx.int(), inv.int())
            B = fx.shape[0]
            fx = _PermuteTokensFn.apply(fx.reshape(B, H * W, C), hit[0], hit[1]).view(B, H // 2, W // 2, 4 * C)
        else:
            fx = F.pad(fx, (0, 0, 0, W % 2, 0, H % 2))
            fx = self._gather4(fx)
        # norm -> the reduction's input quantiser in one kernel each way when the reduction is a QLinear that offers it
        red = self.reduction
        spec = red.fused_input_quant(tuple(fx.shape)) if _SWIN_FUSE and hasattr(red, "fused_input_quant") else None
        fused = F_ofq.norm_quant(self.norm, spec, fx) if spec is not None else None
        if fused is not None:
            _, pre = fused
            return red(pre[0], pre_quant=pre), info
        return self.reduction(F_ofq.layer_norm(self.norm, fx)), info


def relative_position_index(window_size):
    """pair-wise relative position index inside a window (swin.py:220-231)"""
    ch, cw = torch.arange(window_size[0]), torch.arange(window_size[1])
    coords = torch.stack(torch.meshgrid(ch, cw, indexing="ij")).flatten(1)
    rel = (coords[:, :, None] - coords[:, None, :]).permute(1, 2, 0).contiguous()
    rel[:, :, 0] += window_size[0] - 1
    rel[:, :, 1] += window_size[1] - 1
    rel[:, :, 0] *= 2 * window_size[1] - 1
    return rel.sum(-1).view(-1)


_MASK_CACHE = {}


def shift_attention_mask(pad_H, pad_W, window_size, shift_size, device):
    """(num_windows, N, N) additive mask of the shifted-window scheme: 0 inside a region, -100 across (swin.py:134-151).
    A constant of the geometry: built once per (map size, window, shift, device) instead of ~25 small kernels per block and
    step."""
    key = (pad_H, pad_W, tuple(window_size), tuple(shift_size), str(device))
    m = _MASK_CACHE.get(key)
    if m is None:
        m = _MASK_CACHE[key] = _shift_attention_mask(pad_H, pad_W, window_size, shift_size, device)
    return m


def _shift_attention_mask(pad_H, pad_W, window_size, shift_size, device):
    m = torch.zeros((pad_H, pad_W), device=device)
    hs = ((0, -window_size[0]), (-window_size[0], -shift_size[0]), (-shift_size[0], None))
    ws = ((0, -window_size[1]), (-window_size[1], -shift_size[1]), (-shift_size[1], None))
    count = 0
    for h in hs:
        for w in ws:
            m[h[0]:h[1], w[0]:w[1]] = count
            count += 1
    nw = (pad_H // window_size[0]) * (pad_W // window_size[1])
    m = m.view(pad_H // window_size[0], window_size[0], pad_W // window_size[1], window_size[1])
    m = m.permute(0, 2, 1, 3).reshape(nw, window_size[0] * window_size[1])
    m = m.unsqueeze(1) - m.unsqueeze(2)
    return m.masked_fill(m != 0, float(-100.0)).masked_fill(m == 0, float(0.0))


class _RelPosBiasFn(torch.autograd.Function):
    """table[index] (swin.py:232-238, (2*ws-1)^2 x heads -> N*N x heads).  Autograd's backward of the gather is an
    index_put_ with accumulate (60 us per block on sorted keys); here it is one small matmul with the constant 0/1 matrix
    of the index, deterministic."""

    @staticmethod
    def forward(ctx, table, index, onehot_t):
        ctx.save_for_backward(onehot_t)
        return table.index_select(0, index)

    @staticmethod
    def backward(ctx, g):
        (onehot_t,) = ctx.saved_tensors
        return onehot_t @ g.reshape(onehot_t.shape[1], -1), None, None


_ONEHOT_CACHE = {}


def _rel_bias(table, index):
    key = (index.data_ptr(), index.numel(), table.shape[0], str(table.device))
    hit = _ONEHOT_CACHE.get(key)
    if hit is None or hit[0] is not index:
        idx = index.to(table.device).long()
        oh = torch.zeros(table.shape[0], idx.numel(), device=table.device, dtype=table.dtype)
        oh[idx, torch.arange(idx.numel(), device=table.device)] = 1.0
        hit = _ONEHOT_CACHE[key] = (index, idx, oh)
    return _RelPosBiasFn.apply(table, hit[1], hit[2])


class _PermuteTokensFn(torch.autograd.Function):
    """y[b, i] = x[b, idx[i]] for a PERMUTATION idx of the token axis (inv its inverse): one gather each way instead of the
    roll + view + permute + reshape copy chain of the shifted-window partition; the backward of a permutation is the
    gather with the inverse permutation (no scatter-add)."""

    @staticmethod
    def forward(ctx, x, idx, inv):
        ctx.save_for_backward(idx, inv)
        return _permute(x, idx)

    @staticmethod
    def backward(ctx, g):
        idx, inv = ctx.saved_tensors
        return _permute(g, inv), None, None


def _permute(x, idx32):
    if x.dtype == torch.float32 and x.shape[-1] % 4 == 0:
        return ops.permute_tokens(x.contiguous(), idx32)         # csrc/misc.hip, HBM-bound row gather
    return x.index_select(1, idx32.long())


_PERM_CACHE = {}


class WindowGeometry:
    """pad -> cyclic shift -> window partition of a (B, H, W, C) map and the inverse (swin.py:103-131, :160-170)."""

    def __init__(self, x, window_size, shift_size):
        B, H, W, C = x.shape
        self.B, self.H, self.W, self.C = B, H, W, C
        self.ws = list(window_size)
        pad_r = (self.ws[1] - W % self.ws[1]) % self.ws[1]
        pad_b = (self.ws[0] - H % self.ws[0]) % self.ws[0]
        self.pad = (pad_r, pad_b)
        self.pH, self.pW = H + pad_b, W + pad_r
        ss = list(shift_size)
        if self.ws[0] >= self.pH:
            ss[0] = 0
        if self.ws[1] >= self.pW:
            ss[1] = 0
        self.ss = ss
        self.nW = (self.pH // self.ws[0]) * (self.pW // self.ws[1])
        self.N = self.ws[0] * self.ws[1]

    def _perm(self, device):
        """token permutation of partition() (and its inverse) when nothing is padded: built by pushing the token indices
        through the very same pad / roll / view / permute code path, cached per geometry"""
        key = (self.H, self.W, tuple(self.ws), tuple(self.ss), str(device))
        hit = _PERM_CACHE.get(key)
        if hit is None:
            grid = torch.arange(self.H * self.W, device=device, dtype=torch.float32).view(1, self.H, self.W, 1)
            idx = self._partition_copy(grid, 1).reshape(-1).long()
            inv = torch.empty_like(idx)
            inv[idx] = torch.arange(idx.numel(), device=device)
            hit = _PERM_CACHE[key] = (idx.int(), inv.int())
        return hit

    def partition(self, x):
        if self.pad == (0, 0) and x.is_cuda:
            idx, inv = self._perm(x.device)
            y = _PermuteTokensFn.apply(x.reshape(self.B, self.H * self.W, self.C), idx, inv)
            return y.view(self.B * self.nW, self.N, self.C)
        return self._partition_copy(x, self.B)

    def _partition_copy(self, x, B):
        x = F.pad(x, (0, 0, 0, self.pad[0], 0, self.pad[1]))
        if sum(self.ss) > 0:
            x = torch.roll(x, shifts=(-self.ss[0], -self.ss[1]), dims=(1, 2))
        C = x.shape[-1]
        x = x.view(B, self.pH // self.ws[0], self.ws[0], self.pW // self.ws[1], self.ws[1], C)
        return x.permute(0, 1, 3, 2, 4, 5).reshape(B * self.nW, self.N, C)

    def reverse(self, x):
        if self.pad == (0, 0) and x.is_cuda:
            idx, inv = self._perm(x.device)
            y = _PermuteTokensFn.apply(x.reshape(self.B, self.nW * self.N, self.C), inv, idx)
            return y.view(self.B, self.H, self.W, self.C)
        x = x.view(self.B, self.pH // self.ws[0], self.pW // self.ws[1], self.ws[0], self.ws[1], self.C)
        x = x.permute(0, 1, 3, 2, 4, 5).reshape(self.B, self.pH, self.pW, self.C)
        if sum(self.ss) > 0:
            x = torch.roll(x, shifts=(self.ss[0], self.ss[1]), dims=(1, 2))
        return x[:, :self.H, :self.W, :].contiguous()

    def addend(self, table, index, num_heads):
        """(P, N, N) additive term of the softmax: relative-position bias per head (+ shift mask per window);
        slab index = window * heads + head, P = heads (no shift) or windows * heads."""
        gather = _rel_bias(table, index) if table.is_cuda else table[index]
        bias = gather.view(self.N, self.N, -1).permute(2, 0, 1)                     # (heads, N, N)
        if sum(self.ss) == 0:
            return bias.contiguous()
        mask = shift_attention_mask(self.pH, self.pW, self.ws, self.ss, table.device)  # (nW, N, N)
        return (bias.unsqueeze(0) + mask.unsqueeze(1)).reshape(self.nW * num_heads, self.N, self.N)


class ShiftedWindowAttention(nn.Module):
    def __init__(self, dim, window_size, shift_size, num_heads, qkv_bias=True, proj_bias=True, attention_dropout=0.0,
                 dropout=0.0, qqkkvv=False):
        super().__init__()
        if len(window_size) != 2 or len(shift_size) != 2:
            raise ValueError("window_size and shift_size must be of length 2")
        if qqkkvv:
            raise ValueError("qqkkvv score outputs belong to the unused kd_hard_and_soft 2/3 losses (out of scope)")
        self.dim, self.window_size, self.shift_size, self.num_heads = dim, window_size, shift_size, num_heads
        self.attention_dropout, self.dropout, self.qqkkvv = attention_dropout, dropout, qqkkvv
        self.qkv = nn.Linear(dim, dim * 3, bias=qkv_bias)
        self.proj = nn.Linear(dim, dim, bias=proj_bias)
        self.relative_position_bias_table = nn.Parameter(
            torch.zeros((2 * window_size[0] - 1) * (2 * window_size[1] - 1), num_heads))
        self.register_buffer("relative_position_index", relative_position_index(window_size))
        nn.init.trunc_normal_(self.relative_position_bias_table, std=0.02)

    def forward(self, x):
        g = WindowGeometry(x, self.window_size, self.shift_size)
        xw = g.partition(x)
        Bw, N, C = xw.shape
        H = self.num_heads
        qkv = self.qkv(xw).reshape(Bw, N, 3, H, C // H).permute(2, 0, 3, 1, 4)
        q, k, v = qkv[0], qkv[1], qkv[2]
        attn = (q * (C // H) ** -0.5).matmul(k.transpose(-2, -1))
        add = g.addend(self.relative_position_bias_table, self.relative_position_index, H)
        P = add.shape[0]
        attn = (attn.reshape(Bw * H // P, P, N, N) + add).reshape(Bw, H, N, N)
        attn = F.softmax(attn, dim=-1)
        out = attn.matmul(v).transpose(1, 2).reshape(Bw, N, C)
        return g.reverse(self.proj(out)), None


class SwinTransformerBlock(nn.Module):
    def __init__(self, dim, num_heads, window_size, shift_size, mlp_ratio=4.0, dropout=0.0, qqkkvv=False,
                 attention_dropout=0.0, stochastic_depth_prob=0.0, norm_layer=nn.LayerNorm,
                 attn_layer=ShiftedWindowAttention):
        super().__init__()
        if stochastic_depth_prob > 0:
            raise ValueError("stochastic depth is 0 in every OFQ recipe (drop_path: 0.0)")
        self.norm1 = norm_layer(dim)
        self.attn = attn_layer(dim, window_size, shift_size, num_heads, attention_dropout=attention_dropout,
                               dropout=dropout, qqkkvv=qqkkvv)
        self.qqkkvv = qqkkvv
        self.stochastic_depth = nn.Identity()
        self.norm2 = norm_layer(dim)
        self.mlp = MLP(dim, [int(dim * mlp_ratio), dim], activation_layer=nn.GELU, dropout=dropout)
        for m in self.mlp.modules():
            if isinstance(m, nn.Linear):
                nn.init.xavier_uniform_(m.weight)
                if m.bias is not None:
                    nn.init.normal_(m.bias, std=1e-6)

    def forward(self, x):
        x = x[0]
        y, info = self.attn(F_ofq.layer_norm(self.norm1, x))
        x, n2 = F_ofq.add_layer_norm(self.norm2, x, y)
        x = x + self.mlp(n2)
        return x, info

    def forward_fused(self, x, pending):
        """forward() with the residual add of the PREVIOUS block's MLP output (`pending`, not yet added to x) folded into
        norm1's pass; this block's own MLP output is returned un-added.  Returns (x after the attention residual, info,
        mlp output)."""
        plan = self.attn.fused_window_plan(x) if hasattr(self.attn, "fused_window_plan") else None
        if plan is not None:
            # norm1 (+ the pending residual) -> cyclic shift + window partition -> the attention's input quantiser in ONE pass:
            # the codes leave in window-major order, the residual stream stays in token order; the attention's output stays
            # window-major and norm2's pass reads it through the same permutation (no partition / reverse copies either way)
            g, spec, perm = plan
            fused = F_ofq.norm_quant(self.norm1, spec, x, pending, q_perm=perm, qshape=(g.B * g.nW, g.N, g.C))
            if fused is not None:
                xin, pre = fused
                yw = self.attn.window_forward_pre(g, pre)
                spec2 = self.mlp.fused_input_quant(tuple(xin.shape)) if _SWIN_FUSE and hasattr(self.mlp, "fused_input_quant") else None
                fused2 = F_ofq.norm_quant(self.norm2, spec2, xin, yw, res_perm=perm) if spec2 is not None else None
                if fused2 is not None:
                    x2, pre2 = fused2
                    return x2, None, self.mlp(pre2[0], pre_quant=pre2)
                x2, n2 = F_ofq.add_layer_norm(self.norm2, xin, g.reverse(yw))
                return x2, None, self.mlp(n2)
        if pending is None:
            xin, n1 = x, F_ofq.layer_norm(self.norm1, x)
        else:
            xin, n1 = F_ofq.add_layer_norm(self.norm1, x, pending)
        y, info = self.attn(n1)
        # norm2 -> the input quantiser of its only consumer (the MLP's fc1) in one kernel when the MLP offers it (QMLP_swin)
        spec = self.mlp.fused_input_quant(tuple(xin.shape)) if _SWIN_FUSE and hasattr(self.mlp, "fused_input_quant") else None
        fused = F_ofq.norm_quant(self.norm2, spec, xin, y) if spec is not None else None
        if fused is not None:
            x, pre = fused
            return x, info, self.mlp(pre[0], pre_quant=pre)
        x, n2 = F_ofq.add_layer_norm(self.norm2, xin, y)
        return x, info, self.mlp(n2)


class SwinTransformer(nn.Module):
    def __init__(self, patch_size, embed_dim, depths, num_heads, window_size, mlp_ratio=4.0, dropout=0.0, qqkkvv=False,
                 attention_dropout=0.0, stochastic_depth_prob=0.0, num_classes=1000, norm_layer=None, block=None):
        super().__init__()
        self.num_classes = num_classes
        block = block or SwinTransformerBlock
        norm_layer = norm_layer or partial(nn.LayerNorm, eps=1e-5)
        layers = [nn.Sequential(nn.Conv2d(3, embed_dim, kernel_size=tuple(patch_size), stride=tuple(patch_size)),
                                Permute([0, 2, 3, 1]), norm_layer(embed_dim))]
        for i_stage in range(len(depths)):
            dim = embed_dim * 2 ** i_stage
            stage = [block(dim, num_heads[i_stage], window_size=window_size,
                           shift_size=[0 if i_layer % 2 == 0 else w // 2 for w in window_size], mlp_ratio=mlp_ratio,
                           dropout=dropout, qqkkvv=qqkkvv, attention_dropout=attention_dropout,
                           stochastic_depth_prob=0.0, norm_layer=norm_layer) for i_layer in range(depths[i_stage])]
            layers.append(nn.Sequential(*stage))
            if i_stage < len(depths) - 1:
                layers.append(PatchMerging(dim, norm_layer))
        self.features = nn.Sequential(*layers)
        self.qqkkvv = qqkkvv
        num_features = embed_dim * 2 ** (len(depths) - 1)
        self.norm = norm_layer(num_features)
        self.avgpool = nn.AdaptiveAvgPool2d(1)
        self.head = nn.Linear(num_features, num_classes)
        for m in self.modules():
            if isinstance(m, nn.Linear):
                nn.init.trunc_normal_(m.weight, std=0.02)
                if m.bias is not None:
                    nn.init.zeros_(m.bias)

    def forward_features(self, x):
        stem = self.features[0]
        x = stem[1](stem[0](x))
        x = (F_ofq.layer_norm(stem[2], x), None)               # the HIP LayerNorm (nn.Sequential would call the stock one)
        infos = []
        for blk in self.features[1:]:
            if isinstance(blk, nn.Sequential):
                info, pending, cur = None, None, x[0]
                # the W_qk products (and their StatsQ operands) of a stage's QKR blocks in one batched launch each way, as the
                # DeiT models have it (functional.all_wqk: the blocks of a stage share one shape); round 6
                served = F_ofq.all_wqk([b.attn for b in blk if hasattr(b, "attn")]) if _SWIN_WQK and x[0].is_cuda else []
                try:
                    for b in blk:
                        if hasattr(b, "forward_fused"):     # residual adds ride in the next block's norm1 (same values)
                            cur, info, pending = b.forward_fused(cur, pending)
                        else:
                            if pending is not None:
                                cur, pending = cur + pending, None
                            cur, info = b((cur, None))
                finally:
                    for a_ in served:
                        a_._wqk_pre = None
                if pending is not None:
                    cur = cur + pending
                x = (cur, None)
            else:
                x = blk(x)
                info = x[1]
                x = (x[0], None)
            infos.append(info)
        return x[0], infos

    def forward(self, x):
        x, infos = self.forward_features(x)
        x = F_ofq.layer_norm(self.norm, x).permute(0, 3, 1, 2)
        x = torch.flatten(self.avgpool(x), 1)
        return self.head(x), infos


def swin_t(pretrained=False, **kwargs):
    if pretrained:
        raise RuntimeError("pretrained Swin-T weights need network access; load a checkpoint with load_state_dict")
    kwargs.pop("drop_path", None)
    kwargs.pop("weights", None)
    return SwinTransformer(patch_size=[4, 4], embed_dim=kwargs.pop("embed_dim", 96), depths=kwargs.pop("depths", [2, 2, 6, 2]),
                           num_heads=kwargs.pop("num_heads", [3, 6, 12, 24]), window_size=[7, 7], **kwargs)
