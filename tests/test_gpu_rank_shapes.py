"""GPU parity at the per-rank shapes of the two 8-GPU configurations, run on one GPU:
CLIP 1024 own rows of a global batch of 8192 (row offsets 0, 3072, 7168) and FILIP 256 own A
sequences against 2048 gathered B sequences of 256 tokens.  The global batch is formed here on one
device (what the all-gather delivers on eight) and the rank's kernels are checked against float64
mathematics of the whole batch / the oracle on sampled blocks."""
import pytest
import torch

from oracle import pclip_oracle as O
from routing_ref import decode_image, routed_grads

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def K():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    import protein_clip_amd.kernels as kernels
    return kernels


def test_clip_rank_shape_1024_of_8192(K):
    """Loss share, both log-sum-exp vectors and dX / dY of one rank against float64 on the full
    8192 x 8192 logits (training_utils.py:85-89 of the reference and its autograd)."""
    Bg, Br, d = 8192, 1024, 128
    gen = torch.Generator().manual_seed(0)
    e_pep, e_rec = torch.randn(Bg, d, generator=gen), torch.randn(Bg, d, generator=gen)
    t = torch.tensor(1.0)
    x32, y32 = O.l2norm_scale(e_pep, t).cuda(), O.l2norm_scale(e_rec, t).cuda()
    x, y = x32.double(), y32.double()
    S = x @ y.t()
    lse_r, lse_c = torch.logsumexp(S, 1), torch.logsumexp(S, 0)
    diag = torch.diagonal(S)
    eye = torch.eye(Bg, dtype=torch.float64, device="cuda")
    G = (0.5 / Bg) * (torch.softmax(S, 1) + torch.softmax(S, 0) - 2 * eye)
    dX_all, dY_all = G @ y, G.t() @ x
    lr32, lc32 = lse_r.float().contiguous(), lse_c.float().contiguous()
    total = 0.0
    for off in (0, 3072, 7168):
        sl = slice(off, off + Br)
        xo, yo = x32[sl].contiguous(), y32[sl].contiguous()
        got_r, got_c, part, _ = K.clip_fwd(xo, yo, x32, y32, off)
        torch.testing.assert_close(got_r.double(), lse_r[sl], rtol=1e-6, atol=2e-6)
        torch.testing.assert_close(got_c.double(), lse_c[sl], rtol=1e-6, atol=2e-6)
        want_part = (0.5 / Bg) * (lse_r[sl] + lse_c[sl] - 2 * diag[sl]).sum()
        torch.testing.assert_close(part.double().reshape(()), want_part, rtol=1e-5, atol=1e-7)
        total += float(want_part)
        dX, dY = K.clip_bwd(xo, yo, x32, y32, lr32, lc32, off)
        torch.testing.assert_close(dX.double(), dX_all[sl], rtol=1e-4, atol=1e-9)
        torch.testing.assert_close(dY.double(), dY_all[sl], rtol=1e-4, atol=1e-9)
    assert 0 < total < float(torch.log(torch.tensor(float(Bg))))   # three of eight shares of a loss below ln B


@pytest.mark.parametrize("Br,Bg,off,d", [(130, 130, 0, 128), (200, 470, 100, 96), (129, 257, 128, 64), (384, 1000, 616, 128), (512, 513, 1, 4)])
def test_clip_tall_row_blocks_with_ragged_edges(K, Br, Bg, off, d):
    """The kernels that take 128 own rows per workgroup (Br >= 128), at sizes that are no multiple of any
    tile, widths below 128 and own rows at the very end of the batch: log-sum-exps, loss share, the logits in
    both output modes and dX / dY against float64."""
    gen = torch.Generator().manual_seed(Br + Bg)
    x32 = O.l2norm_scale(torch.randn(Bg, d, generator=gen), torch.tensor(1.3)).cuda()
    y32 = O.l2norm_scale(torch.randn(Bg, d, generator=gen), torch.tensor(1.3)).cuda()
    x, y = x32.double(), y32.double()
    S = x @ y.t()
    lse_r, lse_c = torch.logsumexp(S, 1), torch.logsumexp(S, 0)
    G = (0.5 / Bg) * (torch.softmax(S, 1) + torch.softmax(S, 0) - 2 * torch.eye(Bg, dtype=torch.float64, device="cuda"))
    sl = slice(off, off + Br)
    xo, yo = x32[sl].contiguous(), y32[sl].contiguous()
    for mode in (0, 1, 2):
        got_r, got_c, part, logits = K.clip_fwd(xo, yo, x32, y32, off, logits_mode=mode)
        torch.testing.assert_close(got_r.double(), lse_r[sl], rtol=1e-6, atol=2e-6)
        torch.testing.assert_close(got_c.double(), lse_c[sl], rtol=1e-6, atol=2e-6)
        want_part = (0.5 / Bg) * (lse_r[sl] + lse_c[sl] - 2 * torch.diagonal(S)[sl]).sum()
        torch.testing.assert_close(part.double().reshape(()), want_part, rtol=1e-5, atol=1e-7)
        if mode == 1:
            torch.testing.assert_close(logits.double(), S[sl], rtol=1e-6, atol=2e-6)
        if mode == 2:
            torch.testing.assert_close(logits.double(), S[sl].exp(), rtol=1e-5, atol=1e-6)
    dX, dY = K.clip_bwd(xo, yo, x32, y32, lse_r.float().contiguous(), lse_c.float().contiguous(), off)
    torch.testing.assert_close(dX.double(), (G @ y)[sl], rtol=1e-4, atol=1e-8)
    torch.testing.assert_close(dY.double(), (G.t() @ x)[sl], rtol=1e-4, atol=1e-8)


def test_filip_rank_shape_256_by_2048(K):
    """Scores of sampled row / column blocks against the oracle (a pair's scores depend on that
    pair alone), every chosen partner attains its maximum, the backward equals the routed sum for
    every token of both sides, and the gradient of one whole A sequence and one whole B sequence
    against the oracle's autograd."""
    Br, Bc, L, d = 256, 2048, 256, 128
    gen = torch.Generator().manual_seed(8)
    hA = O.bf16_round(torch.randn(Br, L, d, generator=gen))
    hB = O.bf16_round(torch.randn(Bc, L, d, generator=gen))
    mA = O.right_padded_masks(torch.randint(128, L + 1, (Br,), generator=gen), L)
    mB = O.right_padded_masks(torch.randint(128, L + 1, (Bc,), generator=gen), L)
    tt = torch.tensor(1.0)
    t = tt.reshape(1).cuda()
    gA, gB = torch.randn(Br, Bc, generator=gen) / Bc, torch.randn(Br, Bc, generator=gen) / Bc
    hnA, mwA = K.filip_prep(hA.cuda(), mA.cuda(), True, tsign=t)
    hnB, mwB = K.filip_prep(hB.cuda(), mB.cuda(), True)
    simA, simB, argA, argB = K.filip_sim_fwd(hnA, mwA, L, hnB, mwB, L, t, unit_norm=True)
    # scores: three A rows x three column blocks of the global batch
    rows = [3, 100, 255]
    for j0 in (0, 1000, 2040):
        sA, sB = O.filip_similarity(O.bf16_round(O.filip_normalise(hA[rows])), O.bf16_round(O.filip_normalise(hB[j0:j0 + 8])), mA[rows],
                                    mB[j0:j0 + 8], tt, block=8)
        torch.testing.assert_close(simA[rows][:, j0:j0 + 8].cpu(), sA, rtol=2e-6, atol=2e-6)
        torch.testing.assert_close(simB[rows][:, j0:j0 + 8].cpu(), sB, rtol=2e-6, atol=2e-6)
    # routing: the chosen partner attains the maximum (sampled A rows against every B sequence)
    nA, nB = decode_image(hnA, L), decode_image(hnB, L)
    mAd, mBd = mA.cuda().bool(), mB.cuda().bool()
    for i in rows:
        sim = torch.einsum("ad,jbd->jab", nA[i], nB).masked_fill(~(mAd[i][None, :, None] & mBd[:, None, :]), float("-inf"))
        pickA = torch.gather(sim, 2, argA[i, :, :L, None].long().clamp(0, L - 1)).squeeze(2)   # entries of padding tokens are unspecified
        okA = mAd[i][None, :].expand_as(pickA)
        assert (sim.amax(2) - pickA)[okA].abs().max() <= 2e-6
        pickB = torch.gather(sim, 1, argB[i, :, None, :L].long().clamp(0, L - 1)).squeeze(1)
        assert (sim.amax(1) - pickB)[mBd].abs().max() <= 2e-6
    # backward given the routing: every token of both sides
    dnA, dnB, dt = K.filip_sim_bwd(hnA, mwA, L, hnB, mwB, L, t, simA, simB, argA, argB, gA.cuda(), gB.cuda())
    wantA, wantB = routed_grads(nA, nB, mAd, mBd, argA, argB, gA.cuda(), gB.cuda(), 1.0)
    for got, want in ((dnA[:, :L], wantA), (dnB[:, :L], wantB)):
        err = (got.double() - want).norm(dim=-1)
        assert (err <= 2e-6 * want.norm(dim=-1) + 1e-12).all(), float((err / want.norm(dim=-1).clamp_min(1e-30)).max())
    want_dt = -((gA.cuda().double() * simA.double()).sum() + (gB.cuda().double() * simB.double()).sum())
    torch.testing.assert_close(dt.double().reshape(()), want_dt, rtol=1e-4, atol=1e-6)
    # end to end against the oracle's autograd: one A sequence (2048 pairs), one B sequence (256 pairs)
    dA = K.filip_prep_bwd(hA.cuda(), dnA, normalise=True, tsign=t).cpu()
    dB = K.filip_prep_bwd(hB.cuda(), dnB, normalise=True).cpu()

    def oracle(hA_, hB_, mA_, mB_, gA_, gB_):
        a, b = hA_.clone().requires_grad_(True), hB_.clone().requires_grad_(True)
        na, nb = O.filip_normalise(a), O.filip_normalise(b)
        na = na + (O.bf16_round(na.detach()) - na.detach())
        nb = nb + (O.bf16_round(nb.detach()) - nb.detach())
        for j0 in range(0, hB_.shape[0], 256):   # block-wise: the pairs are independent
            s1, s2 = O.filip_similarity_block(na, nb[j0:j0 + 256], mA_, mB_[j0:j0 + 256], tt)
            ((s1 * gA_[:, j0:j0 + 256]).sum() + (s2 * gB_[:, j0:j0 + 256]).sum()).backward(retain_graph=True)
        return a.grad, b.grad

    da, _ = oracle(hA[7:8], hB, mA[7:8], mB, gA[7:8], gB[7:8])
    rel = ((dA[7:8] - da).norm() / da.norm()).item()
    assert rel < 4e-3, rel   # twice what tools/measure_bwd_outliers.py measured at this shape (profiles/r3_bwd_outliers.txt: 1.8e-3, 2.0e-3)
    _, db = oracle(hA, hB[1500:1501], mA, mB[1500:1501], gA[:, 1500:1501], gB[:, 1500:1501])
    rel = ((dB[1500:1501] - db).norm() / db.norm()).item()
    # One B sequence has 256 pairs only, so a single token routed differently from the oracle's f32 host einsum, or one
    # operand that host and device round to different bf16 values, shows: 2.0e-3 and 7.1e-3 were seen for this very
    # sequence on two boxes (the host's rounding depends on its thread count), 7.95e-3 for single B sequences at the
    # benchmark shape (profiles/r3_bwd_outliers.txt).  Budget: twice the largest.  The exact check is the one above
    # (every token equals the float64 sum of the rows its arg indices route to).
    assert rel < 1.6e-2, rel
