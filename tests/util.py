"""Shared helpers for the test-suite: golden loading and comparison metrics."""
import os

import numpy as np
import torch

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load_golden(name):
    z = np.load(os.path.join(GOLDEN_DIR, name + ".npz"))
    return {k: z[k] for k in z.files}


def group(d, prefix):
    """Sub-dict of keys 'prefix:rest' -> rest."""
    n = len(prefix) + 1
    return {k[n:]: v for k, v in d.items() if k.startswith(prefix + ":")}


def case_names(d):
    return sorted({k.split(":")[0] for k in d if ":" in k})


def params(g, requires_grad=True, device="cpu"):
    """'p:<name>' arrays -> dict of leaf tensors."""
    out = {}
    for k, v in g.items():
        if k.startswith("p:"):
            t = torch.from_numpy(np.ascontiguousarray(v)).to(device)
            if requires_grad and t.dtype.is_floating_point:
                t.requires_grad_(True)
            out[k[2:]] = t
    return out


def T(a, device="cpu"):
    return torch.from_numpy(np.ascontiguousarray(a)).to(device)


def rel_err(a, b):
    """max|a-b| / max|b|: the 'fp32 relative tolerance' of BASELINE.json read on the scale of the tensor (max-abs error
    over max-abs value), NOT element-wise -- an element that is tiny next to its neighbours is compared on their scale.
    tests/test_modules_gpu.py::_elementwise_bad_fraction adds the element-wise reading on the non-tiny elements."""
    a = torch.as_tensor(a, dtype=torch.float64).cpu()
    b = torch.as_tensor(b, dtype=torch.float64).cpu()
    den = b.abs().max().item()
    if den == 0.0:
        return (a - b).abs().max().item()
    return (a - b).abs().max().item() / den
