"""Prints, for the backward parity cases of tests/test_gpu_kernels.py, how far the kernels'
gradients are from the oracle's autograd: share of elements outside rtol 1e-3 and relative L2
error.  The thresholds in the tests are set from these measurements."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import conftest  # noqa: E402
import test_gpu_kernels as T  # noqa: E402
from oracle import pclip_oracle as O  # noqa: E402
import protein_clip_amd.kernels as K  # noqa: E402


def report(tag, got, want, atol):
    a, b = got.detach().cpu().double().reshape(want.shape), want.double()
    bad = ((a - b).abs() > atol + 1e-3 * b.abs()).double().mean().item()
    rel = ((a - b).norm() / b.norm().clamp_min(1e-30)).item()
    print(f"{tag:60s} outside {bad:8.4%}   rel L2 {rel:.3e}", flush=True)


def main():
    dev = T.dev
    gen = torch.Generator().manual_seed(21)
    for key, c in conftest.load_cases("filip_small.npz").items():
        pre = key.endswith("bf16n")
        hA, hB, mA, mB, t = c["hA"], c["hB"], c["maskA"], c["maskB"], c["t"]
        B = hA.shape[0]
        gA, gB = torch.randn(B, B, generator=gen), torch.randn(B, B, generator=gen)
        sA, sB, dA_ref, dB_ref, dt_ref = T._oracle_filip_grads(hA, hB, mA, mB, t, not pre, gA, gB)
        hnA, mwA, hnB, mwB, simA, simB, argA, argB = T._filip_fwd(K, hA, hB, mA, mB, t, normalise=not pre)
        dnA, dnB, dt = K.filip_sim_bwd(hnA, mwA, hA.shape[1], hnB, mwB, hB.shape[1], dev(t.reshape(1)), simA, simB, argA, argB, dev(gA), dev(gB))
        report(key + " dA", K.filip_prep_bwd(dev(hA), dnA, normalise=not pre, tsign=dev(t.reshape(1))), dA_ref, 1e-6)
        report(key + " dB", K.filip_prep_bwd(dev(hB), dnB, normalise=not pre), dB_ref, 1e-6)
    # benchmark size, as test_filip_benchmark_size_backward_rows_against_oracle
    gen = torch.Generator().manual_seed(5)
    B, L, d = 256, 256, 128
    hA = O.bf16_round(torch.randn(B, L, d, generator=gen))
    hB = O.bf16_round(torch.randn(B, L, d, generator=gen))
    mA = O.right_padded_masks(torch.randint(100, L + 1, (B,), generator=gen), L)
    mB = O.right_padded_masks(torch.randint(100, L + 1, (B,), generator=gen), L)
    tt = torch.tensor(0.07)
    gA, gB = torch.randn(B, B, generator=gen) / B, torch.randn(B, B, generator=gen) / B
    t = dev(tt.reshape(1))
    hnA, mwA = K.filip_prep(dev(hA), dev(mA), True, tsign=t)
    hnB, mwB = K.filip_prep(dev(hB), dev(mB), True)
    simA, simB, argA, argB = K.filip_sim_fwd(hnA, mwA, L, hnB, mwB, L, t, unit_norm=True)
    dnA, dnB, dt = K.filip_sim_bwd(hnA, mwA, L, hnB, mwB, L, t, simA, simB, argA, argB, dev(gA), dev(gB))
    dA = K.filip_prep_bwd(dev(hA), dnA, normalise=True, tsign=t).cpu()
    dB = K.filip_prep_bwd(dev(hB), dnB, normalise=True).cpu()
    for i in (3, 200):
        _, _, da, _, _ = T._oracle_filip_grads(hA[i:i + 1], hB, mA[i:i + 1], mB, tt, True, gA[i:i + 1], gB[i:i + 1])
        report(f"B=L=256 t=0.07 A sequence {i}", dA[i:i + 1], da, 1e-7)
    for j in (0, 131):
        _, _, _, db, _ = T._oracle_filip_grads(hA, hB[j:j + 1], mA, mB[j:j + 1], tt, True, gA[:, j:j + 1], gB[:, j:j + 1])
        report(f"B=L=256 t=0.07 B sequence {j}", dB[j:j + 1], db, 1e-7)


if __name__ == "__main__":
    main()
