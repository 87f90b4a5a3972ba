"""Largest deviation of the FILIP score kernels from the oracle (same bf16-rounded operands)
over the golden cases and one large random ragged problem."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import pclip_oracle as O  # noqa: E402
from protein_clip_amd import kernels as K  # noqa: E402


def run(hA, hB, mA, mB, t):
    dev = torch.device("cuda:0")
    tt = torch.tensor([float(t)], device=dev)
    hnA, mwA = K.filip_prep(hA.to(dev), mA.to(torch.uint8).to(dev), True, tsign=tt)
    hnB, mwB = K.filip_prep(hB.to(dev), mB.to(torch.uint8).to(dev), True)
    simA, simB, _, _ = K.filip_sim_fwd(hnA, mwA, hA.shape[1], hnB, mwB, hB.shape[1], tt, want_arg=True, unit_norm=True)
    na, nb = O.bf16_round(O.filip_normalise(hA)), O.bf16_round(O.filip_normalise(hB))
    sA, sB = O.filip_similarity(na, nb, mA, mB, torch.tensor(float(t)))
    ok = torch.isfinite(sA) & torch.isfinite(sB)
    dA, dB = (simA.cpu() - sA)[ok].double(), (simB.cpu() - sB)[ok].double()
    print(f"   signed error: mean {dA.mean().item():+.2e} / {dB.mean().item():+.2e}, std {dA.std().item():.2e} / {dB.std().item():.2e}, "
          f"mean score {sA[ok].mean().item():.3f}")
    return max(dA.abs().max().item(), dB.abs().max().item()) * abs(float(t))


def main():
    gen = torch.Generator().manual_seed(9)
    worst = 0.0
    for (B, TA, TB) in [(8, 48, 40), (32, 256, 256), (64, 200, 131)]:
        hA, hB = torch.randn(B, TA, 128, generator=gen), torch.randn(B, TB, 128, generator=gen)
        mA = O.right_padded_masks(torch.randint(1, TA + 1, (B,), generator=gen), TA)
        mB = O.right_padded_masks(torch.randint(1, TB + 1, (B,), generator=gen), TB)
        e = run(hA, hB, mA, mB, 1.0)
        print(f"B={B} TA={TA} TB={TB}: max |score - oracle| = {e:.3e} (unit temperature)")
        worst = max(worst, e)
    print(f"worst {worst:.3e}")


if __name__ == "__main__":
    main()
