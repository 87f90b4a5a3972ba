"""Profiling driver: runs only the FILIP score kernel (benchmark shape) a few times, so a
rocprofv3 pass over it is short.  Usage: python tools/run_filip_fwd.py [reps] [unit|generic]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from protein_clip_amd import kernels as K  # noqa: E402

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 5
unit = (sys.argv[2] if len(sys.argv) > 2 else "unit") == "unit"
full = len(sys.argv) > 3 and sys.argv[3] == "full"
want_arg = not (len(sys.argv) > 4 and sys.argv[4] == "noarg")
B, L, D = 256, 256, 128
g = torch.Generator().manual_seed(0)
hA = torch.randn(B, L, D, generator=g).to(torch.bfloat16).float().cuda()
hB = torch.randn(B, L, D, generator=g).to(torch.bfloat16).float().cuda()
lA = torch.randint(128, L + 1, (B,), generator=g)
lB = torch.randint(128, L + 1, (B,), generator=g)
if full:
    lA[:] = L
    lB[:] = L
ar = torch.arange(L)[None]
mA = (ar < lA[:, None]).to(torch.uint8).cuda()
mB = (ar < lB[:, None]).to(torch.uint8).cuda()
t = torch.tensor([1.0], device="cuda")
hnA, mwA = K.filip_prep(hA, mA, True, tsign=t)
hnB, mwB = K.filip_prep(hB, mB, True)
torch.cuda.synchronize()
ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
for i in range(reps):
    ev[i][0].record()
    out = K.filip_sim_fwd(hnA, mwA, L, hnB, mwB, L, t, want_arg=want_arg, unit_norm=unit)
    ev[i][1].record()
torch.cuda.synchronize()
ms = [a.elapsed_time(b) for a, b in ev]
flops = 2.0 * D * float(lA.sum()) * float(lB.sum())
print("filip_sim ms:", ["%.3f" % m for m in ms], "TFLOP/s(best): %.1f" % (flops / (min(ms) * 1e-3) / 1e12))
