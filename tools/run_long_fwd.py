#!/usr/bin/env python3
"""Profiling driver: the FILIP scores of a batch of long chains, a few times (for rocprofv3 --kernel-trace):
    python3 tools/run_long_fwd.py B T [full] [reps]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from protein_clip_amd import kernels as K  # noqa: E402

B, T = int(sys.argv[1]), int(sys.argv[2])
full = len(sys.argv) > 3 and sys.argv[3] == "full"
reps = int(sys.argv[4]) if len(sys.argv) > 4 else 40
dev = torch.device("cuda", 0)
t = torch.tensor([1.0], device=dev)
g = torch.Generator().manual_seed(T)
hA = torch.randn(B, T, 128, generator=g).to(dev)
hB = torch.randn(B, T, 128, generator=g).to(dev)
lo = T if full else T // 2
mA = (torch.arange(T)[None] < torch.randint(lo, T + 1, (B, 1), generator=g)).to(torch.uint8).to(dev)
mB = (torch.arange(T)[None] < torch.randint(lo, T + 1, (B, 1), generator=g)).to(torch.uint8).to(dev)
hnA, mwA = K.filip_prep(hA, mA, True, tsign=t)
hnB, mwB = K.filip_prep(hB, mB, True)
ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
for a, b in ev:
    a.record()
    K.filip_sim_fwd(hnA, mwA, T, hnB, mwB, T, t, unit_norm=True)
    b.record()
torch.cuda.synchronize()
ms = sorted(a.elapsed_time(b) for a, b in ev)[len(ev) // 2]
flops = 2.0 * 128 * float(mA.sum()) * float(mB.sum())
print(f"B={B} T={T} {'full' if full else 'ragged'}: scores {ms:.4f} ms (median of {reps}, events around the call), {flops / ms / 1e9:.0f} TFLOP/s = "
      f"{flops / ms / 1e9 / 2500:.1%} of the bf16 peak", flush=True)
