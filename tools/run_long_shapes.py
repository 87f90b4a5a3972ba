#!/usr/bin/env python3
"""Times the FILIP score kernel and its backward on batches of long chains (general kernel):
    python3 tools/run_long_shapes.py [B T ...]   (default 16 1000  8 2000)"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from protein_clip_amd import kernels as K  # noqa: E402

args = [int(a) for a in sys.argv[1:]] or [16, 1000, 8, 2000]
dev = torch.device("cuda", 0)
t = torch.tensor([1.0], device=dev)
for B, T in zip(args[0::2], args[1::2]):
    g = torch.Generator().manual_seed(T)
    hA = torch.randn(B, T, 128, generator=g).to(dev)
    hB = torch.randn(B, T, 128, generator=g).to(dev)
    mA = (torch.arange(T)[None] < torch.randint(T // 2, T + 1, (B, 1), generator=g)).to(torch.uint8).to(dev)
    mB = (torch.arange(T)[None] < torch.randint(T // 2, T + 1, (B, 1), generator=g)).to(torch.uint8).to(dev)
    hnA, mwA = K.filip_prep(hA, mA, True, tsign=t)
    hnB, mwB = K.filip_prep(hB, mB, True)
    out = K.filip_sim_fwd(hnA, mwA, T, hnB, mwB, T, t, unit_norm=True)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5):
        out = K.filip_sim_fwd(hnA, mwA, T, hnB, mwB, T, t, unit_norm=True)
    torch.cuda.synchronize()
    ms_f = (time.perf_counter() - t0) / 5 * 1e3
    gA = torch.randn(B, B, device=dev) / B
    K.filip_sim_bwd(hnA, mwA, T, hnB, mwB, T, t, *out, gA, gA)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5):
        K.filip_sim_bwd(hnA, mwA, T, hnB, mwB, T, t, *out, gA, gA)
    torch.cuda.synchronize()
    ms_b = (time.perf_counter() - t0) / 5 * 1e3
    flops = 2.0 * 128 * float(mA.sum()) * float(mB.sum())
    print(f"B={B} T={T}: scores {ms_f:.3f} ms ({flops / ms_f / 1e9:.0f} TFLOP/s), backward {ms_b:.3f} ms", flush=True)
