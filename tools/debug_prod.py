#!/usr/bin/env python3
"""Where does a production-dimension case (tests/golden/prodcases.py) differ from the oracle?  Per-token error map."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, ROOT + "/tests", ROOT + "/tests/golden", ROOT + "/oracle"):
    sys.path.insert(0, p)
import numpy as np, torch
import prodcases as PC
from util import load_golden, group, T
from test_oracle_golden import _ofq_namespace, prod_oracle_forward
name = sys.argv[1] if len(sys.argv) > 1 else "swin_qkr_384"
g = group(load_golden("g10_prod"), name)
q, x, oi = PC.build(name, _ofq_namespace())
q.cuda().train()
with torch.no_grad():
    q(x.cuda())
sd = q.state_dict()
for k, v in g.items():
    if k.startswith("p:"):
        sd[k[2:]] = T(v).cuda()
q.load_state_dict(sd)
with torch.no_grad():
    y = q(x.cuda())
    y = y[oi] if oi is not None else y
p = {k: v.detach().cpu() for k, v in q.state_dict().items()}
with torch.no_grad():
    yo = prod_oracle_forward(name)(x, p)
err = (y.cpu() - yo).abs().reshape(-1, y.shape[-1])
tok = err.max(1).values / yo.abs().max()
print("tokens:", tok.numel(), " tokens with err > 1e-4:", int((tok > 1e-4).sum()), " > 1e-3:", int((tok > 1e-3).sum()))
idx = torch.nonzero(tok > 1e-4).reshape(-1)[:20]
for i in idx:
    e = err[i]
    print(" token", int(i), "max rel %.3e" % float(tok[i]), " channels off > 1e-4: %d" % int((e / yo.abs().max() > 1e-4).sum()))
# repeatability and grad-mode dependence
outs = []
for mode in ("nograd", "grad", "grad", "nograd"):
    if mode == "nograd":
        with torch.no_grad():
            yy = q(x.cuda())
    else:
        xg = x.cuda().requires_grad_(True)
        yy = q(xg)
    yy = (yy[oi] if oi is not None else yy).detach().cpu()
    e = (yy - yo).abs().reshape(-1, yy.shape[-1]).max(1).values / yo.abs().max()
    print(mode, "tokens off > 1e-4:", int((e > 1e-4).sum()), "max %.3e" % float(e.max()), "first bad tokens", torch.nonzero(e > 1e-4).reshape(-1)[:8].tolist())
    outs.append(yy)
print("HIP vs golden:", PC.compare(y, g, "y"))
print("oracle(HIP params) vs golden:", PC.compare(yo, g, "y"))
q2, x2, _ = PC.build(name, _ofq_namespace())
p2 = {k: v.detach().clone() for k, v in q2.state_dict().items()}
for k, v in g.items():
    if k.startswith("p:"):
        p2[k[2:]] = T(v).clone()
with torch.no_grad():
    yo2 = prod_oracle_forward(name)(x2, p2)
print("oracle(CPU ctor params) vs golden:", PC.compare(yo2, g, "y"))
for k in p2:
    a, b = p2[k].double(), p[k].double()
    if a.shape != b.shape or float((a - b).abs().max()) > 0:
        print("param differs:", k, tuple(a.shape), tuple(b.shape), float((a - b).abs().max()) if a.shape == b.shape else "")
print("x same:", torch.equal(x, x2))
