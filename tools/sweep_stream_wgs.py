"""Timing sweep of the FILIP score kernel over the number of workgroups it aims at (benchmark
shape, ragged or full length); the scores do not depend on it.  Usage:
python3 tools/sweep_stream_wgs.py [ragged|full] [wgs ...]"""
import os
import statistics
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from protein_clip_amd import kernels as K  # noqa: E402

full = len(sys.argv) > 1 and sys.argv[1] == "full"
targets = [int(a) for a in sys.argv[2:]] or [512, 640, 768, 896, 1024]
B, L, D = 256, 256, 128
g = torch.Generator().manual_seed(0)
torch.randn(B, D, generator=g), torch.randn(B, D, generator=g)
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


def run(wgs, reps):
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
    for i in range(reps):
        ev[i][0].record()
        K.filip_sim_fwd(hnA, mwA, L, hnB, mwB, L, t, want_arg=True, unit_norm=True, stream_wgs=wgs)
        ev[i][1].record()
    torch.cuda.synchronize()
    return statistics.median(a.elapsed_time(b) for a, b in ev)


run(512, 60)   # clocks settle
for rnd in range(3):
    print("round", rnd, " ".join("%d: %.4f" % (w, run(w, 40)) for w in targets), flush=True)
