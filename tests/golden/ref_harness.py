"""Container-only harness: import the upstream reference (nbasyl/OFQ, mounted read-only at
/root/reference) on a CPU-only box so golden vectors can be generated from the reference itself.

The reference cannot be imported as-is here: it needs timm 0.5.4, torchvision 0.15 and
tkinter (`from turtle import forward`), and every LSQ `init_from` hard-codes device="cuda".  This file
pre-seeds `sys.modules` with minimal stand-ins for those *third-party* names and redirects "cuda" to
CPU.  None of the stand-ins performs arithmetic on the fake-quant path except `PatchEmbed`
(Conv2d(k=s=patch) + flatten + transpose, the timm 0.5.4 definition) and `trunc_normal_` (init only).

Nothing here travels to the GPU box's run-time path: tests/, bench.py and smoke() never import this
module; only `make_golden.py` does, and only where /root/reference exists.
"""
import sys
import types
import math

import torch
import torch.nn as nn

REFERENCE_ROOT = "/root/reference"


def _mod(name):
    m = types.ModuleType(name)
    sys.modules[name] = m
    return m


def _to_2tuple(x):
    return tuple(x) if isinstance(x, (tuple, list)) else (x, x)


class _PatchEmbed(nn.Module):
    # timm 0.5.4 layers/patch_embed.py semantics: conv(k=s=patch) -> flatten(2) -> transpose(1, 2)
    def __init__(self, img_size=224, patch_size=16, in_chans=3, embed_dim=768, norm_layer=None, flatten=True):
        super().__init__()
        img_size = _to_2tuple(img_size)
        patch_size = _to_2tuple(patch_size)
        self.img_size = img_size
        self.patch_size = patch_size
        self.grid_size = (img_size[0] // patch_size[0], img_size[1] // patch_size[1])
        self.num_patches = self.grid_size[0] * self.grid_size[1]
        self.flatten = flatten
        self.proj = nn.Conv2d(in_chans, embed_dim, kernel_size=patch_size, stride=patch_size)
        self.norm = norm_layer(embed_dim) if norm_layer else nn.Identity()

    def forward(self, x):
        x = self.proj(x)
        if self.flatten:
            x = x.flatten(2).transpose(1, 2)
        return self.norm(x)


class _DropPath(nn.Module):
    def __init__(self, drop_prob=0.0):
        super().__init__()
        self.drop_prob = drop_prob

    def forward(self, x):
        assert self.drop_prob == 0.0 or not self.training
        return x


def _trunc_normal_(tensor, mean=0.0, std=1.0, a=-2.0, b=2.0):
    return nn.init.trunc_normal_(tensor, mean=mean, std=std, a=a, b=b)


def _lecun_normal_(tensor):
    fan_in = tensor.shape[1] if tensor.ndim > 1 else tensor.shape[0]
    return nn.init.trunc_normal_(tensor, std=math.sqrt(1.0 / fan_in) / 0.87962566103423978)


def install_stubs():
    if "timm" in sys.modules and getattr(sys.modules["timm"], "_ofq_stub", False):
        return
    timm = _mod("timm")
    timm._ofq_stub = True
    data = _mod("timm.data")
    data.IMAGENET_DEFAULT_MEAN = (0.485, 0.456, 0.406)
    data.IMAGENET_DEFAULT_STD = (0.229, 0.224, 0.225)
    data.IMAGENET_INCEPTION_MEAN = (0.5, 0.5, 0.5)
    data.IMAGENET_INCEPTION_STD = (0.5, 0.5, 0.5)
    models = _mod("timm.models")
    helpers = _mod("timm.models.helpers")
    helpers.build_model_with_cfg = None
    helpers.named_apply = None
    helpers.adapt_input_conv = None
    layers = _mod("timm.models.layers")
    layers.PatchEmbed = _PatchEmbed
    layers.DropPath = _DropPath
    layers.trunc_normal_ = _trunc_normal_
    layers.lecun_normal_ = _lecun_normal_
    layers.to_2tuple = _to_2tuple
    registry = _mod("timm.models.registry")
    registry.register_model = lambda f: f
    loss = _mod("timm.loss")

    class SoftTargetCrossEntropy(nn.Module):
        def forward(self, x, target):
            return torch.sum(-target * torch.log_softmax(x, dim=-1), dim=-1).mean()

    loss.SoftTargetCrossEntropy = SoftTargetCrossEntropy
    timm.data, timm.models, timm.loss = data, models, loss
    models.helpers, models.layers, models.registry = helpers, layers, registry

    turtle = _mod("turtle")
    turtle.forward = None

    # torchvision: only names, never called on the DeiT path
    tv = _mod("torchvision")
    ops = _mod("torchvision.ops")
    misc = _mod("torchvision.ops.misc")

    class MLP(nn.Sequential):
        def __init__(self, in_channels, hidden_channels, norm_layer=None, activation_layer=nn.ReLU,
                     inplace=None, bias=True, dropout=0.0):
            layers_ = []
            in_dim = in_channels
            for hidden_dim in hidden_channels[:-1]:
                layers_.append(nn.Linear(in_dim, hidden_dim, bias=bias))
                layers_.append(activation_layer())
                layers_.append(nn.Dropout(dropout))
                in_dim = hidden_dim
            layers_.append(nn.Linear(in_dim, hidden_channels[-1], bias=bias))
            layers_.append(nn.Dropout(dropout))
            super().__init__(*layers_)

    class Permute(nn.Module):
        def __init__(self, dims):
            super().__init__()
            self.dims = dims

        def forward(self, x):
            return torch.permute(x, self.dims)

    misc.MLP, misc.Permute = MLP, Permute
    sd = _mod("torchvision.ops.stochastic_depth")

    class StochasticDepth(nn.Module):
        def __init__(self, p, mode):
            super().__init__()
            self.p = p

        def forward(self, x):
            assert self.p == 0.0 or not self.training
            return x

    sd.StochasticDepth = StochasticDepth
    tr = _mod("torchvision.transforms")
    presets = _mod("torchvision.transforms._presets")
    presets.ImageClassification = object
    presets.InterpolationMode = types.SimpleNamespace(BICUBIC="bicubic")
    tvu = _mod("torchvision.utils")
    tvu._log_api_usage_once = lambda *a, **k: None
    tvm = _mod("torchvision.models")
    api = _mod("torchvision.models._api")

    class Weights:
        def __init__(self, *a, **k):
            pass

    class WeightsEnum:
        pass

    api.Weights, api.WeightsEnum = Weights, WeightsEnum
    meta = _mod("torchvision.models._meta")
    meta._IMAGENET_CATEGORIES = []
    tvmu = _mod("torchvision.models._utils")
    tvmu._ovewrite_named_param = lambda *a, **k: None
    tv.ops, tv.transforms, tv.utils, tv.models = ops, tr, tvu, tvm
    ops.misc, ops.stochastic_depth = misc, sd


_shimmed = False


def install_cuda_shim():
    """Redirect device="cuda" allocations and .cuda() calls to CPU (lsq.py:57-69 etc.)."""
    global _shimmed
    if _shimmed:
        return
    _shimmed = True
    _zeros = torch.zeros
    _zeros_like = torch.zeros_like

    def zeros(*a, **k):
        if str(k.get("device", "")).startswith("cuda"):
            k["device"] = "cpu"
        return _zeros(*a, **k)

    torch.zeros = zeros
    torch.Tensor.cuda = lambda self, *a, **k: self
    nn.Module.cuda = lambda self, *a, **k: self


def import_reference():
    """Returns the imported `src` package of the reference."""
    install_stubs()
    install_cuda_shim()
    if REFERENCE_ROOT not in sys.path:
        sys.path.insert(0, REFERENCE_ROOT)
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        import src  # noqa: F401
    return sys.modules["src"]


if __name__ == "__main__":
    s = import_reference()
    print("reference imported:", [n for n in dir(s) if n.startswith("Q") or n.startswith("Lsq")][:12])
