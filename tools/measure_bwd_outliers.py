"""How far the FILIP backward is from the oracle's autograd at the benchmark shape (256 x 256) and at one rank's shape
of the 8-GPU configuration (256 x 2048), and WHY: for whole sequences, the share of gradient elements outside
rtol 1e-3, the relative L2 error, the number of token rows with such elements, how many of those rows are touched by a
routing difference (the kernel's arg-max partner is not the float64 arg-max of the same bf16 operands, on the row or on
the column side), and the largest gap between the two candidates' similarities over all routing differences (a
near-tie: closer than the f32 rounding of two different summation orders).  The budgets in
tests/test_gpu_kernels.py and tests/test_gpu_rank_shapes.py are set from this output (kept under profiles/)."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import test_gpu_kernels as T  # noqa: E402
from oracle import pclip_oracle as O  # noqa: E402
import protein_clip_amd.kernels as K  # noqa: E402

dev = T.dev


def oracle_grads(hA, hB, mA, mB, tt, gA, gB):
    """Autograd of the oracle on the normalised, bf16-rounded tokens, block-wise over B (the pairs are independent)."""
    a, b = hA.clone().requires_grad_(True), hB.clone().requires_grad_(True)
    na, nb = O.filip_normalise(a), O.filip_normalise(b)
    na = na + (O.bf16_round(na.detach()) - na.detach())
    nb = nb + (O.bf16_round(nb.detach()) - nb.detach())
    for j0 in range(0, hB.shape[0], 256):
        s1, s2 = O.filip_similarity_block(na, nb[j0:j0 + 256], mA, mB[j0:j0 + 256], tt)
        ((s1 * gA[:, j0:j0 + 256]).sum() + (s2 * gB[:, j0:j0 + 256]).sum()).backward(retain_graph=True)
    return a.grad, b.grad


def routing_differences(n_own, m_own, n_oth, m_oth, arg_own, arg_oth, f32_cpu=False):
    """One own sequence against all partner sequences.  The reference arg-max is that of float64 products on the device,
    or — ``f32_cpu`` — of the f32 einsum on the host that the oracle itself routes by (in blocks of 256 partners, as
    oracle_grads calls it); the gap between two candidates is always measured in float64.
    n_own [T,128], m_own [T]; n_oth [Bo,T,128], m_oth [Bo,T]; arg_own [Bo,T] (own token -> partner token),
    arg_oth [Bo,T] (partner token -> own token).  Returns (own tokens touched by a difference [T] bool, largest gap)."""
    # entries of padding tokens are unspecified (possibly never written): keep every index inside the tables
    T_own, T_oth = n_own.shape[0], n_oth.shape[1]
    arg_own = arg_own.long().clamp(0, T_oth - 1)
    arg_oth = arg_oth.long().clamp(0, T_own - 1)
    S = torch.einsum("ad,jbd->jab", n_own.double(), n_oth.double())                      # [Bo, T own, T partner]
    neg = torch.finfo(torch.float64).min
    S = S.masked_fill(~(m_own.bool()[None, :, None] & m_oth.bool()[:, None, :]), neg)
    touched = torch.zeros(n_own.shape[0], dtype=torch.bool, device=S.device)
    gap = 0.0
    if f32_cpu:
        a, refs, refcs = n_own.float().cpu()[None], [], []
        ok = (m_own.bool()[None, :, None] & m_oth.bool()[:, None, :]).cpu()
        for j0 in range(0, n_oth.shape[0], 256):
            s32 = torch.einsum("iad,jbd->ijab", a, n_oth[j0:j0 + 256].float().cpu())[0]
            s32 = torch.where(ok[j0:j0 + 256], s32, torch.full((), torch.finfo(torch.float32).min))
            refs.append(s32.argmax(dim=2))
            refcs.append(s32.argmax(dim=1))
        ref32, refc32 = torch.cat(refs).to(S.device), torch.cat(refcs).to(S.device)
    # own token -> partner token
    best, ref = S.max(dim=2)
    if f32_cpu:
        ref = ref32
        best = torch.gather(S, 2, ref[:, :, None])[:, :, 0]
    got = torch.gather(S, 2, arg_own.long()[:, :, None])[:, :, 0]
    valid = m_own.bool()[None, :] & (m_oth.sum(1) > 0)[:, None]
    diff = valid & (arg_own.long() != ref)
    touched |= diff.any(0)
    if diff.any():
        gap = max(gap, float((best - got)[diff].abs().max()))
    # partner token -> own token
    bestc, refc = S.max(dim=1)                                                            # [Bo, T partner]
    if f32_cpu:
        refc = refc32
        bestc = torch.gather(S, 1, refc[:, None, :])[:, 0, :]
    gotc = torch.gather(S, 1, arg_oth.long()[:, None, :])[:, 0, :]
    validc = m_oth.bool() & bool(m_own.sum() > 0)
    diffc = validc & (arg_oth.long() != refc)
    if diffc.any():
        gap = max(gap, float((bestc - gotc)[diffc].abs().max()))
        touched[arg_oth.long()[diffc]] = True
        touched[refc[diffc]] = True
    return touched, gap


def report(tag, got, want, touched64, gap64, touched32, gap32, flips, atol=1e-7):
    a, b = got.detach().cpu().double().reshape(want.shape), want.double()
    bad = (a - b).abs() > atol + 1e-3 * b.abs()
    rel = ((a - b).norm() / b.norm().clamp_min(1e-30)).item()
    bad_rows = bad.reshape(-1, bad.shape[-1]).any(-1)
    t64, t32, fl = touched64.cpu(), touched32.cpu(), flips.cpu()
    print(f"{tag:28s} outside {bad.double().mean().item():8.4%}  rel L2 {rel:.2e}  token rows with outliers {int(bad_rows.sum()):2d}: "
          f"{int((bad_rows & t32).sum()):2d} routed differently from the oracle (f32 host einsum; {int(t32.sum())} such rows, gap <= {gap32:.1e}), "
          f"{int((bad_rows & fl & ~t32).sum()):2d} more with an operand that the device and the host rounded to different bf16 values "
          f"({int(fl.sum())} such rows); against the float64 arg-max the kernel differs on {int(t64.sum())} rows (gap <= {gap64:.1e})", flush=True)


def run(Br, Bc, seed, seqs_a, seqs_b, tt):
    gen = torch.Generator().manual_seed(seed)
    L, d = 256, 128
    hA = O.bf16_round(torch.randn(Br, L, d, generator=gen))
    hB = O.bf16_round(torch.randn(Bc, L, d, generator=gen))
    mA = O.right_padded_masks(torch.randint(100, L + 1, (Br,), generator=gen), L)
    mB = O.right_padded_masks(torch.randint(100, L + 1, (Bc,), generator=gen), L)
    gA, gB = torch.randn(Br, Bc, generator=gen) / Bc, torch.randn(Br, Bc, generator=gen) / Bc
    t = dev(tt.reshape(1))
    hnA, mwA = K.filip_prep(dev(hA), dev(mA), True, tsign=t)
    hnB, mwB = K.filip_prep(dev(hB), dev(mB), True)
    simA, simB, argA, argB = K.filip_sim_fwd(hnA, mwA, L, hnB, mwB, L, t, unit_norm=True)
    dnA, dnB, _ = K.filip_sim_bwd(hnA, mwA, L, hnB, mwB, L, t, simA, simB, argA, argB, dev(gA), dev(gB))
    dA = K.filip_prep_bwd(dev(hA), dnA, normalise=True, tsign=t).cpu()
    dB = K.filip_prep_bwd(dev(hB), dnB, normalise=True).cpu()
    nA = hnA.view(torch.bfloat16).float()[:, :L] * (-1.0 if tt < 0 else 1.0)
    nB = hnB.view(torch.bfloat16).float()[:, :L]
    mAd, mBd = dev(mA), dev(mB)
    # the oracle's operands: normalised and rounded on the host (an ulp of difference in the norm flips a bf16 rounding now and then)
    flipA = (dev(O.bf16_round(O.filip_normalise(hA))) != nA).any(-1) & mAd.bool()
    flipB = (dev(O.bf16_round(O.filip_normalise(hB))) != nB).any(-1) & mBd.bool()
    print(f"-- {Br} x {Bc} sequences, L = 256, t = {float(tt):.2f} (similarities are scaled by 1/t = {1 / float(tt):.1f}); tokens whose bf16 operand "
          f"differs between device and host normalisation: A {int(flipA.sum())}, B {int(flipB.sum())}")
    for i in seqs_a:
        da, _ = oracle_grads(hA[i:i + 1], hB, mA[i:i + 1], mB, tt, gA[i:i + 1], gB[i:i + 1])
        t64, g64 = routing_differences(nA[i], mAd[i], nB, mBd, argA[i, :, :L], argB[i, :, :L])
        t32, g32 = routing_differences(nA[i], mAd[i], nB, mBd, argA[i, :, :L], argB[i, :, :L], f32_cpu=True)
        report(f"A sequence {i} ({Bc} pairs)", dA[i:i + 1], da, t64, g64, t32, g32, flipA[i])
    for j in seqs_b:
        _, db = oracle_grads(hA, hB[j:j + 1], mA, mB[j:j + 1], tt, gA[:, j:j + 1], gB[:, j:j + 1])
        t64, g64 = routing_differences(nB[j], mBd[j], nA, mAd, argB[:, j, :L], argA[:, j, :L])
        t32, g32 = routing_differences(nB[j], mBd[j], nA, mAd, argB[:, j, :L], argA[:, j, :L], f32_cpu=True)
        report(f"B sequence {j} ({Br} pairs)", dB[j:j + 1], db, t64, g64, t32, g32, flipB[j])


if __name__ == "__main__":
    run(256, 256, 5, (3, 200), (0, 131), torch.tensor(0.07))     # test_filip_benchmark_size_backward_rows_against_oracle
    run(256, 256, 5, (3, 200), (0, 131), torch.tensor(1.0))
    run(256, 2048, 7, (7,), (1500,), torch.tensor(1.0))          # one rank of the 8-GPU configuration
