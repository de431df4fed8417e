"""Model surgery — drop-in for src/quantization/modules/utils.py:21-282 (`replace_module_by_qmodule_deit`,
the QMODULE_MAPPINGS tables, get/set_module_by_name).  Same arguments, same name-list driven replacement;
patch embedding and both heads are forced to W8A8 LSQ exactly like the reference (:126-156)."""
import torch

from .qlinear import LSQ_QConv2d, QLinear, QMLP, LSQ_QLinear4head
from .attention import QAttention, QAttention_qkreparam, QAttention_qkreparam_4_cga
from ...deit_vision_transformer import Attention as deit_attention, Mlp
from ...swin import ShiftedWindowAttention, MLP as swin_MLP
from .swin_attention_and_mlp import QAttention_swin, QMLP_swin, QAttention_swin_qkreparam, QAttention_swin_qkreparam_4_cga

QMODULE_MAPPINGS = {torch.nn.Linear: QLinear, deit_attention: QAttention, Mlp: QMLP}
# 0: QAttention_qkreparam, 1: QAttention_qkreparam_4_cga                        (modules/utils.py:27-39)
QMODULE_MAPPINGS_QK_REPARAM = [
    {torch.nn.Linear: QLinear, deit_attention: QAttention_qkreparam, Mlp: QMLP},
    {torch.nn.Linear: QLinear, deit_attention: QAttention_qkreparam_4_cga, Mlp: QMLP},
]


def get_module_by_name(model, module_name):
    module = model
    for name in module_name.split("."):
        module = getattr(module, name)
    return module


def set_module_by_name(model, module_name, module):
    names = module_name.split(".")
    parent = get_module_by_name(model, ".".join(names[:-1])) if len(names) > 1 else model
    setattr(parent, names[-1], module)


_W8A8 = dict(weight_bits=8, input_bits=8, weight_channelwise=True, input_channelwise=True, weight_quant_method='lsq',
             input_quant_method='lsq', aq_learnable=True, wq_learnable=True)


def replace_module_by_qmodule_deit(model, qconfigs, pretrained_initialized=False, qk_reparam=False, qk_reparam_type=0,
                                   boundaryRange=0.005):
    first = qconfigs[list(qconfigs.keys())[0]]
    if first["weight"]["mode"] == 'lsq' and first["act"]["mode"] == 'lsq':
        raise ValueError("the LSQ-weights baseline (LSQ_w_and_act_*) is not part of any shipped OFQ recipe")
    mapping = QMODULE_MAPPINGS_QK_REPARAM[qk_reparam_type] if qk_reparam else QMODULE_MAPPINGS
    for name, cfg in qconfigs.items():
        module = get_module_by_name(model, name)
        if name == "patch_embed.proj":
            qmodule = LSQ_QConv2d(m=module, act_layer=cfg["act_layer"], pretrained_initialized=pretrained_initialized,
                                  **_W8A8)
        elif name == "head" or name == "head_dist":
            qmodule = LSQ_QLinear4head(m=module, symmetric=True, act_layer=cfg["act_layer"],
                                       pretrained_initialized=pretrained_initialized, **_W8A8)
        else:
            if type(module) not in mapping:
                raise KeyError("no quantised counterpart for %s (%s)" % (name, type(module).__name__))
            extra = {"boundaryRange": boundaryRange} if (qk_reparam and qk_reparam_type == 1) else {}
            qmodule = mapping[type(module)](
                m=module, weight_bits=cfg["weight"]['bit'], input_bits=cfg["act"]['bit'],
                weight_channelwise=cfg["weight"]["per_channel"], input_channelwise=cfg["act"]["per_channel"],
                weight_quant_method=cfg["weight"]["mode"], input_quant_method=cfg["act"]["mode"],
                aq_learnable=cfg["act"]["learnable"], wq_learnable=cfg["weight"]["learnable"],
                act_layer=cfg["act_layer"], pretrained_initialized=pretrained_initialized, **extra)
        set_module_by_name(model, name, qmodule)
    return model


# ---- Swin (modules/utils.py:286-413) ---------------------------------------------------------------------------
QMODULE_MAPPINGS_SWIN = {torch.nn.Linear: QLinear, ShiftedWindowAttention: QAttention_swin, swin_MLP: QMLP_swin}
QMODULE_MAPPINGS_QK_REPARAM_SWIN = [
    {torch.nn.Linear: QLinear, ShiftedWindowAttention: QAttention_swin_qkreparam, swin_MLP: QMLP_swin},
    {torch.nn.Linear: QLinear, ShiftedWindowAttention: QAttention_swin_qkreparam_4_cga, swin_MLP: QMLP_swin},
]


def replace_module_by_qmodule_swin(model, qconfigs, pretrained_initialized=False, qk_reparam=False, qk_reparam_type=0,
                                   boundaryRange=0.005):
    """Name-list driven surgery for Swin: `features.0.0` (4x4 patch conv) and `head` are forced to W8A8 LSQ, attention /
    MLP / `reduction` linears follow the per-module config (modules/utils.py:305-413)."""
    mapping = QMODULE_MAPPINGS_QK_REPARAM_SWIN[qk_reparam_type] if qk_reparam else QMODULE_MAPPINGS_SWIN
    for name, cfg in qconfigs.items():
        module = get_module_by_name(model, name)
        if name == "features.0.0":
            qmodule = LSQ_QConv2d(m=module, act_layer=cfg["act_layer"], pretrained_initialized=pretrained_initialized,
                                  **_W8A8)
        elif name == "head":
            qmodule = LSQ_QLinear4head(m=module, symmetric=True, act_layer=cfg["act_layer"],
                                       pretrained_initialized=pretrained_initialized, **_W8A8)
        else:
            if type(module) not in mapping:
                raise KeyError("no quantised counterpart for %s (%s)" % (name, type(module).__name__))
            qmodule = mapping[type(module)](
                m=module, weight_bits=cfg["weight"]['bit'], input_bits=cfg["act"]['bit'],
                weight_channelwise=cfg["weight"]["per_channel"], input_channelwise=cfg["act"]["per_channel"],
                weight_quant_method=cfg["weight"]["mode"], input_quant_method=cfg["act"]["mode"],
                aq_learnable=cfg["act"]["learnable"], wq_learnable=cfg["weight"]["learnable"],
                act_layer=cfg["act_layer"], pretrained_initialized=pretrained_initialized)
        set_module_by_name(model, name, qmodule)
    return model
