"""Profiling driver: the FILIP hot-path step at the benchmark shape (B=256, L=256, d=128,
valid lengths U[128,256]) a few times and nothing else, so that a rocprofv3 counter pass over
it is short.  Usage: python3 tools/run_filip_step.py [reps] [full]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from protein_clip_amd import kernels as K  # noqa: E402

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 3
full = len(sys.argv) > 2 and sys.argv[2] == "full"
B, L, D = 256, 256, 128
g = torch.Generator().manual_seed(0)
torch.randn(B, D, generator=g), torch.randn(B, D, generator=g)      # same stream position as bench.py
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
step = K.FilipStep(B, B, L, L, D, "cuda:0")
torch.cuda.synchronize()
ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
t0 = torch.cuda.Event(enable_timing=True)
t1 = torch.cuda.Event(enable_timing=True)
for i in range(reps):
    if i == reps - 1:
        t0.record()
    out = step.run(hA, hB, mA, mB, t, sim_events=ev[i])
t1.record()
torch.cuda.synchronize()
ms = [a.elapsed_time(b) for a, b in ev]
flops = 2.0 * D * float(lA.sum()) * float(lB.sum())
print("filip_sim ms:", ["%.3f" % m for m in ms], "TFLOP/s(best): %.1f" % (flops / (min(ms) * 1e-3) / 1e12),
      "step ms (last): %.3f" % t0.elapsed_time(t1), "loss %.6f" % ((out[0] + out[1]) / 2).item())
