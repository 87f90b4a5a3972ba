"""Distribution of |kernel - reference| over all 2 x 65,536 scores of the benchmark-size bf16-exact fixture
(tests/golden/filip_b256_l256_bf16n.npz), and where the worst pair's error comes from: per token, the
kernel's chosen partner evaluated in float64 against the float64 maximum."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import pclip_oracle as O  # noqa: E402
from protein_clip_amd import kernels as K  # noqa: E402


def main():
    z = np.load(os.path.join(ROOT, "tests", "golden", "filip_b256_l256_bf16n.npz"))
    B, L, d = 256, 256, 128
    gen = torch.Generator().manual_seed(0)
    hA = O.bf16_round(torch.randn(B, L, d, generator=gen))
    hB = O.bf16_round(torch.randn(B, L, d, generator=gen))
    lA = torch.randint(128, 257, (B,), generator=gen)
    lB = torch.randint(128, 257, (B,), generator=gen)
    mA, mB = O.right_padded_masks(lA, L), O.right_padded_masks(lB, L)
    # the expression oracle/gen_golden.py fed to the reference (torch's norm)
    nA = (hA / hA.norm(dim=-1, keepdim=True)).to(torch.bfloat16).float()
    nB = (hB / hB.norm(dim=-1, keepdim=True)).to(torch.bfloat16).float()
    dev = torch.device("cuda:0")
    t = torch.tensor([1.0], device=dev)
    hnA, mwA = K.filip_prep(nA.to(dev), mA.to(dev), False, tsign=t)
    hnB, mwB = K.filip_prep(nB.to(dev), mB.to(dev), False)
    simA, simB, argA, argB = K.filip_sim_fwd(hnA, mwA, L, hnB, mwB, L, t, want_arg=True, unit_norm=True)
    for name, got, want in (("sim_A", simA.cpu(), torch.from_numpy(z["sim_A"])), ("sim_B", simB.cpu(), torch.from_numpy(z["sim_B"]))):
        e = (got - want).double()
        q = torch.quantile(e.abs().flatten(), torch.tensor([0.5, 0.99, 0.9999], dtype=torch.float64))
        i, j = divmod(int(e.abs().argmax()), B)
        print(f"{name}: mean {e.mean():+.2e} std {e.std():.2e} |e| median {q[0]:.2e} p99 {q[1]:.2e} p99.99 {q[2]:.2e} max {e.abs().max():.3e} at ({i},{j}); "
              f"over 2e-6: {(e.abs() > 2e-6).sum().item()}, over 1e-6: {(e.abs() > 1e-6).sum().item()}")
        # the worst pair in float64
        a, b = nA[i, :lA[i]].double(), nB[j, :lB[j]].double()
        S = a @ b.t()
        if name == "sim_A":
            exact = S.max(1).values.mean().item()
            arg = argA[i, j, :lA[i]].cpu().long()
            chosen = S[torch.arange(lA[i]), arg]
            gap = S.max(1).values - chosen
        else:
            exact = S.max(0).values.mean().item()
            arg = argB[i, j, :lB[j]].cpu().long()
            chosen = S[arg, torch.arange(lB[j])]
            gap = S.max(0).values - chosen
        print(f"   pair ({i},{j}): float64 {exact:.9f} reference(f32) {want[i, j].item():.9f} kernel {got[i, j].item():.9f}; "
              f"kernel's partners evaluated in float64: mean {chosen.mean().item():.9f}, largest shortfall to the true maximum {gap.max().item():.2e}")


if __name__ == "__main__":
    main()
