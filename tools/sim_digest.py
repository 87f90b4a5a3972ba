"""Regression digest of the FILIP score kernel: checksums of the scores and of the arg-max indices of
valid tokens for a few shapes (benchmark shape, short ragged sequences, masks with holes), so that
a change meant to leave results bit-identical can be checked against the previous build."""
import hashlib
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from protein_clip_amd import kernels as K  # noqa: E402


def digest(t):
    return hashlib.sha1(t.contiguous().cpu().numpy().tobytes()).hexdigest()[:12]


def run(name, B, Bc, TA, TB, lo, holes, seed):
    g = torch.Generator().manual_seed(seed)
    hA = torch.randn(B, TA, 128, generator=g).to(torch.bfloat16).float().cuda()
    hB = torch.randn(Bc, TB, 128, generator=g).to(torch.bfloat16).float().cuda()
    if holes:
        mA = (torch.rand(B, TA, generator=g) > 0.3); mA[:, 0] = True
        mB = (torch.rand(Bc, TB, generator=g) > 0.3); mB[:, 0] = True
        mB[1] = False                                  # an empty sequence
    else:
        lA = torch.randint(lo, TA + 1, (B,), generator=g)
        lB = torch.randint(lo, TB + 1, (Bc,), generator=g)
        mA = torch.arange(TA)[None] < lA[:, None]
        mB = torch.arange(TB)[None] < lB[:, None]
    mA, mB = mA.to(torch.uint8).cuda(), mB.to(torch.uint8).cuda()
    t = torch.tensor([0.9], device="cuda")
    hnA, mwA = K.filip_prep(hA, mA, True, tsign=t)
    hnB, mwB = K.filip_prep(hB, mB, True)
    sA, sB, aA, aB = K.filip_sim_fwd(hnA, mwA, TA, hnB, mwB, TB, t, want_arg=True, unit_norm=True)
    torch.cuda.synchronize()
    vA = mA.bool()[:, None, :].expand(B, Bc, TA) & (mB.sum(1) > 0)[None, :, None]
    vB = mB.bool()[None, :, :].expand(B, Bc, TB) & (mA.sum(1) > 0)[:, None, None]
    aA = torch.where(vA, aA[..., :TA].to(torch.int32) & 0xffff, torch.zeros((), dtype=torch.int32, device="cuda"))
    aB = torch.where(vB, aB[..., :TB].to(torch.int32) & 0xffff, torch.zeros((), dtype=torch.int32, device="cuda"))
    print(f"{name}: simA {digest(sA)} simB {digest(sB)} argA {digest(aA)} argB {digest(aB)}", flush=True)


run("bench 256x256 L256", 256, 256, 256, 256, 128, False, 0)
run("short 40x24", 40, 24, 70, 96, 3, False, 1)
run("holes 33x17", 33, 17, 100, 64, 0, True, 2)
run("rank 64x512", 64, 512, 256, 256, 128, False, 3)
