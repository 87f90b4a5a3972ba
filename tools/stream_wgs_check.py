"""Prints one checksum of the FILIP scores and arg-max tables of a ragged problem per workgroup
target of the streaming kernel given on the command line (default: 2 40 512 5000); the lines
must all be equal."""
import hashlib
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from protein_clip_amd import kernels as K  # noqa: E402


def checksum(wgs):
    gen = torch.Generator().manual_seed(3)
    Br, Bc, TA, TB, d = 37, 83, 200, 256, 128
    dev = torch.device("cuda:0")
    hA = torch.randn(Br, TA, d, generator=gen).to(dev)
    hB = torch.randn(Bc, TB, d, generator=gen).to(dev)
    lA = torch.randint(1, TA + 1, (Br,), generator=gen)
    lB = torch.randint(0, TB + 1, (Bc,), generator=gen)
    mA = (torch.arange(TA)[None, :] < lA[:, None]).to(torch.uint8).to(dev)
    mB = (torch.arange(TB)[None, :] < lB[:, None]).to(torch.uint8).to(dev)
    t = torch.tensor([0.5], device=dev)
    hnA, mwA = K.filip_prep(hA, mA, True, tsign=t)
    hnB, mwB = K.filip_prep(hB, mB, True)
    simA, simB, argA, argB = K.filip_sim_fwd(hnA, mwA, TA, hnB, mwB, TB, t, want_arg=True, unit_norm=True, stream_wgs=wgs)
    # arg entries of padding tokens are unspecified: zero them through the masks before hashing
    TAp, TBp = argA.shape[-1], argB.shape[-1]
    vA = torch.zeros(Br, TAp, dtype=torch.bool, device=dev)
    vA[:, :TA] = mA.bool()
    vB = torch.zeros(Bc, TBp, dtype=torch.bool, device=dev)
    vB[:, :TB] = mB.bool()
    live = (lB > 0).to(dev)
    argA = torch.where(vA[:, None, :] & live[None, :, None], argA, torch.zeros_like(argA))
    argB = torch.where(vB[None, :, :], argB, torch.zeros_like(argB))
    h = hashlib.sha256()
    for x in (simA, simB, argA, argB):
        h.update(x.cpu().numpy().tobytes())
    return h.hexdigest()


if __name__ == "__main__":
    for wgs in [int(a) for a in sys.argv[1:]] or [2, 40, 512, 5000]:
        print("checksum", wgs, checksum(wgs))
