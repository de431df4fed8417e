#!/usr/bin/env python3
"""Where does the hand-off of a cut tile spend its time?  Needs a -DNTSK_PHASE_PROBE -DNTSK_CLOCK_PROBE build of the library in
OFQ_HIP_LIB: every workgroup of one stream-K launch (G = 256, cut tiles) stamps s_memrealtime at: end of its head piece (0),
publish done (1), end of its last k-step as an owner (2), first partial's flag seen (3), partials added (4), everything stored (5)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from ofq_amd import ops
M = 128 * 198
G = int(os.environ.get("WGS", "256"))
torch.manual_seed(0)
for (o, c) in [(2304, 384), (1536, 384), (384, 1536)]:
    dy = torch.randn(M, o, device="cuda") * 1e-3
    qw = (2 * torch.randint(-2, 2, (o, c), device="cuda") + 1).to(torch.int8)
    wT = ops.codes_transpose_bf16(qw)
    ks = torch.rand(o, device="cuda") + 0.5
    out = torch.empty(M, c, device="cuda")
    for _ in range(5):
        ops.qgemm_bf16s_nt_sk([(dy, wT, ks, 0.25)], out, wgs=G)
    torch.cuda.synchronize()
    fl = ops._sk_workspace(out.device)[:32768].view(torch.int32)[4200:4200 + 8 * G].view(G, 8).cpu().float() / 100.0   # us
    pub = fl[:, 1] > 0
    own = fl[:, 4] > 0
    print("N=%d K=%d G=%d: kernel end (max stamp 5) %.1f us, median %.1f us" % (c, o, G, fl[:, 5].max(), fl[:, 5].median()))
    if pub.any():
        d = (fl[pub, 1] - fl[pub, 0])
        print("   publishers %3d: publish (stores + drain + flag) median %.1f us, max %.1f; head piece ends at median %.1f us, max %.1f"
              % (int(pub.sum()), d.median(), d.max(), fl[pub, 0].median(), fl[pub, 0].max()))
    if own.any():
        wait = fl[own, 3] - fl[own, 2]
        gat = fl[own, 4] - fl[own, 3]
        sto = fl[own, 5] - fl[own, 4]
        print("   owners     %3d: last k-step ends at median %.1f us (max %.1f); wait for the first flag median %.1f us, max %.1f; "
              "acquire + add partials median %.1f us, max %.1f; store median %.1f us, max %.1f"
              % (int(own.sum()), fl[own, 2].median(), fl[own, 2].max(), wait.median(), wait.max(), gat.median(), gat.max(), sto.median(), sto.max()))
        late = own & (fl[:, 5] > fl[:, 5].median() + 5)
        print("   workgroups finishing > 5 us after the median: %d" % int(late.sum()))
