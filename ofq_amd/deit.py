"""DistilledVisionTransformer and the two DeiT factories — counterpart of src/deit.py:19-105
(198 tokens: cls + dist + 196 patches; LayerNorm eps 1e-6; returns ((cls, dist), attn_list) in training
mode and the averaged logits in eval mode)."""
from functools import partial

import torch
import torch.nn as nn

from .deit_vision_transformer import VisionTransformer, trunc_normal_

__all__ = ['deit_tiny_distilled_patch16_224', 'deit_small_distilled_patch16_224', 'DistilledVisionTransformer',
           'create_model']


class DistilledVisionTransformer(VisionTransformer):
    def __init__(self, *args, **kwargs):
        super().__init__(*args, **kwargs)
        self.dist_token = nn.Parameter(torch.zeros(1, 1, self.embed_dim))
        num_patches = self.patch_embed.num_patches
        self.pos_embed = nn.Parameter(torch.zeros(1, num_patches + 2, self.embed_dim))
        self.head_dist = nn.Linear(self.embed_dim, self.num_classes) if self.num_classes > 0 else nn.Identity()
        trunc_normal_(self.dist_token, std=.02)
        trunc_normal_(self.pos_embed, std=.02)
        self.head_dist.apply(self._init_weights)


def _deit(embed_dim, num_heads, pretrained=False, **kwargs):
    if pretrained:
        raise RuntimeError("pretrained DeiT weights need network access; load a checkpoint with load_state_dict")
    kwargs.pop("pretrained_cfg", None)
    return DistilledVisionTransformer(patch_size=16, embed_dim=embed_dim, depth=kwargs.pop("depth", 12),
                                      num_heads=num_heads, mlp_ratio=4, qkv_bias=True,
                                      norm_layer=partial(nn.LayerNorm, eps=1e-6), act_layer=nn.GELU, **kwargs)


def deit_tiny_distilled_patch16_224(pretrained=False, **kwargs):
    return _deit(192, 3, pretrained, **kwargs)


def deit_small_distilled_patch16_224(pretrained=False, **kwargs):
    return _deit(384, 6, pretrained, **kwargs)


def _swin_t(**kw):
    from .swin import swin_t
    return swin_t(**kw)


_REGISTRY = {f.__name__: f for f in (deit_tiny_distilled_patch16_224, deit_small_distilled_patch16_224)}
_REGISTRY["swin_t"] = _swin_t


def create_model(name, **kwargs):
    """Stand-in for timm.create_model (train.py:503) restricted to the model families on the hot path."""
    if name not in _REGISTRY:
        raise ValueError("unknown model %r (available: %s)" % (name, sorted(_REGISTRY)))
    if kwargs.pop("drop_rate", 0.0):
        raise ValueError("drop_rate is 0 in every OFQ recipe; the quantised modules do not apply dropout")
    return _REGISTRY[name](**kwargs)
