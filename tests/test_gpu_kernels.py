"""GPU parity tests of the HIP kernels, called through the C ABI (ctypes) and checked against
the CPU oracle and the reference-generated golden vectors.  Run with ``-m gpu`` on an MI355X."""
import math

import pytest
import torch

from conftest import head_case_params, load_flat
from oracle import pclip_oracle as O
from routing_ref import decode_image, routed_grads

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def K():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    import protein_clip_amd.kernels as kernels
    from protein_clip_amd import _lib
    assert _lib.load().pclip_check_device(0) == 0, _lib.load().pclip_last_error()
    return kernels


def dev(t):
    return t.cuda().contiguous()


def close(a, b, rtol, atol):
    torch.testing.assert_close(a.detach().cpu().float().reshape(b.shape), b.float(), rtol=rtol, atol=atol)


def mostly_close(a, b, rtol, atol, max_bad_frac=0.0, max_rel_l2=None):
    """Elementwise closeness with a budget of outliers (arg-max near-ties route a gradient row
    to a different, equally good partner) and an overall relative L2 bound."""
    a = a.detach().cpu().double().reshape(b.shape)
    b = b.double()
    bad = (a - b).abs() > (atol + rtol * b.abs())
    frac = bad.double().mean().item()
    assert frac <= max_bad_frac, f"{frac:.4%} of elements outside tolerance (budget {max_bad_frac:.4%})"
    if max_rel_l2 is not None:
        rel = ((a - b).norm() / b.norm().clamp_min(1e-30)).item()
        assert rel <= max_rel_l2, f"relative L2 error {rel:.3e} > {max_rel_l2:.1e}"


def test_clip_loss_grad_from_tower_outputs(K, clip_cases):
    """pclip_clip_loss_grad (loss and both gradients from x, y as the towers return them) against the
    reference's loss and autograd on the golden cases, and against the forward + backward entry points at
    sizes on both sides of the one-launch limit."""
    for key, c in clip_cases.items():
        x = c["x"].clone().requires_grad_(True)
        y = c["y"].clone().requires_grad_(True)
        O.clip_contrastive_loss(x, y.t()).backward()
        loss, dx, dy = K.clip_loss_grad(dev(c["x"]), dev(c["y"]))
        close(loss, c["loss"].reshape(1), 1e-5, 1e-6)
        close(dx, x.grad, 1e-4, 1e-7)
        close(dy, y.grad, 1e-4, 1e-7)
    g = torch.Generator().manual_seed(11)
    for B, d in [(256, 128), (512, 128), (544, 128), (100, 64)]:
        x = dev(O.l2norm_scale(torch.randn(B, d, generator=g), torch.tensor(1.0)))
        y = dev(O.l2norm_scale(torch.randn(B, d, generator=g), torch.tensor(1.0)))
        loss, dx, dy = K.clip_loss_grad(x, y)
        lr, lc, want_loss, _ = K.clip_fwd(x, y, x, y, 0)
        want_dx, want_dy = K.clip_bwd(x, y, x, y, lr, lc, 0)
        torch.testing.assert_close(loss, want_loss, rtol=1e-6, atol=1e-7)
        torch.testing.assert_close(dx, want_dx, rtol=1e-4, atol=1e-9)
        torch.testing.assert_close(dy, want_dy, rtol=1e-4, atol=1e-9)


# ------------------------------------------------------------------------------------------
# embedding head: pooling -> Linear/ReLU/LayerNorm/Dropout blocks -> Linear -> normalise, scale
# ------------------------------------------------------------------------------------------
def test_head_tail_matches_the_reference_encoder(K, head_cases):
    """The reference's Encoder after the per-residue network (models.py:48-51), values and every
    gradient, with the dropout masks the reference drew."""
    for key, c in head_cases.items():
        params, grads, keep, scale = head_case_params(c)
        tokens, t = dev(c["tokens"]), dev(c["t"].reshape(1))
        pooled, inv, m8 = K.masked_mean_fwd(tokens, dev(c["mask"]))
        close(pooled, c["pooled"], 1e-5, 1e-6)
        P = [dev(p) for p in params]
        kp = None if keep is None else dev(keep)
        out, save = K.head_tail_fwd(pooled, P, kp, scale, 1e-5, t)
        close(out, c["out"], 1e-4, 1e-6)
        d_pooled, d_params, d_t = K.head_tail_bwd(pooled, P, kp, scale, 1e-5, t, save, dev(c["d_out"]))
        close(d_pooled, c["d_pooled"], 2e-4, 2e-6)
        close(d_t, c["d_t"].reshape(1), 2e-4, 1e-5)
        for g, want in zip(d_params, grads):
            close(g, want, 2e-4, 2e-5)
        close(K.masked_mean_bwd(d_pooled, m8, inv, tokens.shape[1]), c["d_tokens"], 2e-4, 2e-6)


@pytest.mark.parametrize("B,d,nblk,p", [(256, 128, 1, 0.1), (1024, 128, 1, 0.1), (7, 128, 4, 0.5), (33, 64, 0, 0.0), (300, 96, 2, 0.0), (1, 4, 1, 0.3)])
def test_head_tail_against_the_oracle(K, B, d, nblk, p):
    """Benchmark batch sizes, the deepest head the kernel takes, a lone Linear, narrow and odd widths:
    against the oracle in float64 (f32 kernel: 1e-5 of the largest magnitude per tensor)."""
    g = torch.Generator().manual_seed(B + d + nblk)
    pooled = torch.randn(B, d, generator=g)
    params = []
    for l in range(nblk):
        params += [torch.randn(d, d, generator=g) / math.sqrt(d), 0.1 * torch.randn(d, generator=g), 1 + 0.2 * torch.randn(d, generator=g),
                   0.1 * torch.randn(d, generator=g)]
    params += [torch.randn(d, d, generator=g) / math.sqrt(d), 0.1 * torch.randn(d, generator=g)]
    keep = (torch.rand(nblk, B, d, generator=g) >= p).to(torch.uint8) if p > 0 else None
    scale = 1.0 / (1.0 - p)
    t = torch.tensor(0.9)
    d_out = torch.randn(B, d, generator=g)
    leaves = [v.double().requires_grad_(True) for v in [pooled] + params]
    t64 = t.double().requires_grad_(True)
    want = O.head_tail(leaves[0], leaves[1:], keep, scale, 1e-5, t64)
    (want * d_out.double()).sum().backward()

    P = [dev(v) for v in params]
    kp = None if keep is None else dev(keep)
    out, save = K.head_tail_fwd(dev(pooled), P, kp, scale, 1e-5, dev(t.reshape(1)))
    dt = torch.full((1,), 0.25, device="cuda")                      # accumulated into
    d_pooled, d_params, dt2 = K.head_tail_bwd(dev(pooled), P, kp, scale, 1e-5, dev(t.reshape(1)), save, dev(d_out), dt=dt)
    assert dt2 is dt

    def near(a, b, rel=1e-5):
        b = b.detach()
        err = (a.detach().cpu().double().reshape(b.shape) - b).abs().max().item()
        assert err <= rel * max(b.abs().max().item(), 1e-3), f"max error {err:.3e} against magnitude {b.abs().max().item():.3e}"
    # a LayerNorm over four values, most of them clipped to zero, is ill-conditioned: looser there
    grad_rel = 2e-5 if d >= 64 else 2e-3
    near(out, want)
    near(d_pooled, leaves[0].grad, grad_rel)
    for got, leaf in zip(d_params, leaves[1:]):
        near(got, leaf.grad, grad_rel)
    # dt is a sum over rows of terms of either sign: the f32 error scales with the sum of magnitudes
    terms = 0.5 * (want.detach() * d_out.double()).sum(1)
    assert abs((dt - 0.25).item() - t64.grad.item()) <= 1e-6 * terms.abs().sum().item() + 1e-7


def test_head_tail_rejects_what_it_cannot_run(K):
    from protein_clip_amd._lib import PclipError
    x = torch.randn(8, 130, device="cuda")
    with pytest.raises(PclipError):
        K.head_tail_fwd(x, [torch.randn(130, 130, device="cuda"), torch.randn(130, device="cuda")], None, 1.0, 1e-5, torch.ones(1, device="cuda"))
    x = torch.randn(8, 128, device="cuda")
    blocks = [torch.randn(128, 128, device="cuda"), torch.randn(128, device="cuda"), torch.ones(128, device="cuda"), torch.zeros(128, device="cuda")] * 5
    with pytest.raises(PclipError):
        K.head_tail_fwd(x, blocks + [torch.randn(128, 128, device="cuda"), torch.randn(128, device="cuda")], None, 1.0, 1e-5, torch.ones(1, device="cuda"))
    with pytest.raises(PclipError):   # CPU tensors are not taken
        K.head_tail_fwd(x.cpu(), [torch.randn(128, 128), torch.randn(128)], None, 1.0, 1e-5, torch.ones(1))


# ------------------------------------------------------------------------------------------
# L2 normalise + scale
# ------------------------------------------------------------------------------------------
def test_l2norm_scale_matches_golden(K, clip_cases):
    for key, c in clip_cases.items():
        x = K.l2norm_scale_fwd(dev(c["e_pep"]), dev(c["t"].reshape(1)))
        close(x, c["x"], 1e-6, 1e-7)


def test_l2norm_scale_bwd_matches_autograd(K):
    g = torch.Generator().manual_seed(7)
    for B, d in [(5, 8), (33, 128), (64, 100), (3, 1000)]:
        e = torch.randn(B, d, generator=g)
        dx = torch.randn(B, d, generator=g)
        t = torch.tensor(0.8)
        e_ = e.clone().requires_grad_(True)
        t_ = t.clone().requires_grad_(True)
        (O.l2norm_scale(e_, t_) * dx).sum().backward()
        de, dt = K.l2norm_scale_bwd(dev(e), dev(t.reshape(1)), dev(dx))
        close(de, e_.grad, 1e-4, 1e-6)
        close(dt, t_.grad.reshape(1), 1e-4, 1e-5)


# ------------------------------------------------------------------------------------------
# CLIP
# ------------------------------------------------------------------------------------------
def test_clip_fwd_matches_golden(K, clip_cases):
    for key, c in clip_cases.items():
        x, y = dev(c["x"]), dev(c["y"])
        lse_r, lse_c, loss, logits = K.clip_fwd(x, y, x, y, 0, logits_mode=1)
        close(loss, c["loss"].reshape(1), 1e-5, 1e-6)
        close(logits, c["logits"], 1e-5, 2e-6)
        close(lse_r, torch.logsumexp(c["logits"], 1), 1e-5, 2e-6)
        close(lse_c, torch.logsumexp(c["logits"], 0), 1e-5, 2e-6)
        _, _, _, e_logits = K.clip_fwd(x, y, x, y, 0, logits_mode=2)
        close(e_logits, c["exp_logits"], 1e-5, 1e-6)


def test_clip_bwd_matches_oracle_autograd(K, clip_cases):
    for key, c in clip_cases.items():
        x = c["x"].clone().requires_grad_(True)
        y = c["y"].clone().requires_grad_(True)
        (O.clip_contrastive_loss(x, y.t()) * 3.0).backward()
        xd, yd = dev(c["x"]), dev(c["y"])
        lse_r, lse_c, _, _ = K.clip_fwd(xd, yd, xd, yd)
        dX, dY = K.clip_bwd(xd, yd, xd, yd, lse_r, lse_c, 0, g=torch.tensor([3.0]))
        close(dX, x.grad, 1e-4, 1e-7)
        close(dY, y.grad, 1e-4, 1e-7)


def test_clip_step_matches_golden(K, clip_cases):
    for key, c in clip_cases.items():
        loss, dp, dr, dt = K.clip_step(dev(c["e_pep"]), dev(c["e_rec"]), dev(c["t"].reshape(1)))
        close(loss, c["loss"].reshape(1), 1e-5, 1e-6)
        close(dp, c["d_e_pep"], 1e-4, 2e-7)
        close(dr, c["d_e_rec"], 1e-4, 2e-7)
        close(dt, c["d_t"].reshape(1), 1e-4, 2e-6)


def test_clip_benchmark_size_scalars(K):
    g = load_flat("clip_scalars.npz")
    for B in (256, 1024):
        gen = torch.Generator().manual_seed(0)
        e_pep = torch.randn(B, 128, generator=gen)
        e_rec = torch.randn(B, 128, generator=gen)
        loss, dp, dr, dt = K.clip_step(dev(e_pep), dev(e_rec), dev(torch.tensor([1.0])))
        close(loss, g[f"loss_B{B}"].reshape(1), 1e-5, 1e-5)
        ol, odp, odr, odt = O.clip_step_analytic(e_pep.double(), e_rec.double(), torch.tensor(1.0, dtype=torch.float64))
        close(dp, odp.float(), 1e-4, 1e-8)
        close(dr, odr.float(), 1e-4, 1e-8)
        close(dt, odt.float().reshape(1), 1e-4, 1e-6)
        # size-independent property: the gradient of the loss w.r.t. every logit sums to zero per
        # row block, so d e . e = 0 for each row (the normalise Jacobian kills the radial part)
        radial = (dp.cpu() * e_pep).sum(-1).abs().max().item()
        assert radial < 1e-6


def test_clip_random_shapes_against_oracle(K):
    """Seeded sweep of the fused CLIP step over batch sizes around the 32-row tile edges and
    embedding widths from 4 to 128, temperatures 0, 1, 2 and a peaked case."""
    gen = torch.Generator().manual_seed(77)
    pick = lambda xs: xs[int(torch.randint(len(xs), (1,), generator=gen))]
    for trial in range(16):
        B, d = pick([1, 2, 5, 31, 32, 33, 100, 257]), pick([4, 8, 64, 100, 128])
        t = torch.tensor(pick([0.0, 1.0, 2.0]), dtype=torch.float64)
        e_pep = torch.randn(B, d, generator=gen)
        e_rec = e_pep + 0.1 * torch.randn(B, d, generator=gen) if trial % 2 else torch.randn(B, d, generator=gen)
        ol, odp, odr, odt = O.clip_step_analytic(e_pep.double(), e_rec.double(), t)
        loss, dp, dr, dt = K.clip_step(dev(e_pep), dev(e_rec), dev(t.float().reshape(1)))
        tag = f"trial {trial}: B={B} d={d} t={float(t)}"
        assert torch.allclose(loss.cpu().double(), ol.reshape(1), rtol=2e-5, atol=2e-6), tag
        assert torch.allclose(dp.cpu().double(), odp, rtol=2e-4, atol=1e-6), tag
        assert torch.allclose(dr.cpu().double(), odr, rtol=2e-4, atol=1e-6), tag
        assert torch.allclose(dt.cpu().double(), odt.reshape(1), rtol=2e-4, atol=5e-6), tag


def test_clip_sharded_rows_equal_single_device(K):
    """Two 'ranks' each owning half the rows reproduce the one-device result exactly as the
    multi-GPU path assembles it: lse vectors concatenated, loss parts summed."""
    gen = torch.Generator().manual_seed(3)
    B, d = 96, 128
    x = O.l2norm_scale(torch.randn(B, d, generator=gen), torch.tensor(1.0))
    y = O.l2norm_scale(torch.randn(B, d, generator=gen), torch.tensor(1.0))
    xd, yd = dev(x), dev(y)
    lse_r, lse_c, loss, _ = K.clip_fwd(xd, yd, xd, yd)
    dX, dY = K.clip_bwd(xd, yd, xd, yd, lse_r, lse_c)
    parts = []
    for r, (lo, hi) in enumerate([(0, 40), (40, 96)]):
        xo, yo = xd[lo:hi].contiguous(), yd[lo:hi].contiguous()
        parts.append((lo, hi, xo, yo) + K.clip_fwd(xo, yo, xd, yd, lo)[:3])
    lr = torch.cat([p[4] for p in parts]); lc = torch.cat([p[5] for p in parts])
    close(lr, lse_r.cpu(), 1e-6, 1e-6)
    close(lc, lse_c.cpu(), 1e-6, 1e-6)
    close(sum(p[6] for p in parts), loss.cpu(), 1e-6, 1e-6)
    for lo, hi, xo, yo, *_ in parts:
        dXo, dYo = K.clip_bwd(xo, yo, xd, yd, lr, lc, lo)
        close(dXo, dX[lo:hi].cpu(), 1e-5, 1e-8)
        close(dYo, dY[lo:hi].cpu(), 1e-5, 1e-8)


def test_clip_rejects_bad_arguments(K):
    from protein_clip_amd._lib import PclipError
    x = torch.randn(8, 130, device="cuda")
    with pytest.raises(PclipError):
        K.clip_fwd(x, x, x, x)            # d not a multiple of 4 / too wide
    with pytest.raises(PclipError):
        K.l2norm_scale_fwd(torch.randn(4, 8), 1.0)   # CPU tensor: no fallback


# ------------------------------------------------------------------------------------------
# row-wise InfoNCE
# ------------------------------------------------------------------------------------------
@pytest.mark.parametrize("use_dcl", [False, True])
def test_rowce_matches_oracle(K, use_dcl):
    gen = torch.Generator().manual_seed(11)
    for B in (3, 16, 200):
        logits = torch.randn(B, B, generator=gen) * 3
        l_ = logits.clone().requires_grad_(True)
        loss_ref = O.row_ce(l_, use_dcl)
        (loss_ref * 0.5).backward()
        lse, loss = K.rowce_fwd(dev(logits), use_dcl=use_dcl)
        close(loss, loss_ref.detach().reshape(1), 1e-5, 1e-6)
        dl = K.rowce_bwd(dev(logits), lse, use_dcl=use_dcl, gscale=0.5)
        close(dl, l_.grad, 1e-4, 1e-8)


def test_rowce_uniform_logits_give_log_B(K):
    lse, loss = K.rowce_fwd(torch.zeros(64, 64, device="cuda"))
    close(loss, torch.tensor([math.log(64.0)]), 1e-6, 1e-6)


@pytest.mark.parametrize("Br,Bc,off", [(5, 5, 0), (64, 256, 128), (256, 256, 0), (33, 2048, 1000)])
def test_filip_loss_equals_the_row_ce_calls_and_the_oracle(K, Br, Bc, off):
    """The two-launch FILIP loss against the four row-CE calls it replaces and against the
    oracle's _compute_loss on the square cases."""
    gen = torch.Generator().manual_seed(Br * 7 + Bc)
    sA, sB = torch.randn(Br, Bc, generator=gen) * 3, torch.randn(Br, Bc, generator=gen) * 3
    dA, dB = dev(sA), dev(sB)
    loss3, gA, gB = K.filip_loss(dA, dB, off, Bc)
    lseA, lossA = K.rowce_fwd(dA, off, Bc, False)
    lseB, lossB = K.rowce_fwd(dB, off, Bc, False)
    close(gA, K.rowce_bwd(dA, lseA, off, Bc, False, gscale=0.5).cpu(), 1e-5, 1e-9)
    close(gB, K.rowce_bwd(dB, lseB, off, Bc, False, gscale=0.5).cpu(), 1e-5, 1e-9)
    close(loss3[1:2], lossA.cpu(), 1e-6, 1e-7)
    close(loss3[2:3], lossB.cpu(), 1e-6, 1e-7)
    close(loss3[0:1], ((lossA + lossB) / 2).cpu(), 1e-6, 1e-7)
    if Br == Bc and off == 0:
        a = sA.clone().requires_grad_(True)
        b = sB.clone().requires_grad_(True)
        ref = O.filip_loss(a, b)
        ref.backward()
        close(loss3[0:1], ref.detach().reshape(1), 1e-5, 1e-6)
        close(gA, a.grad, 1e-5, 1e-8)
        close(gB, b.grad, 1e-5, 1e-8)


def test_masked_mean_matches_the_reference_formula(K):
    """Encoder._masked_mean (modules/models.py:76-90): forward and backward against autograd of
    the same formula, widths with and without a 16-byte row pitch, an empty sequence, int64 masks."""
    from protein_clip_amd import ops
    gen = torch.Generator().manual_seed(12)
    for B, T, d in [(3, 7, 128), (5, 300, 128), (2, 33, 20), (4, 65, 130), (1, 1, 7), (2, 70, 132), (3, 40, 256)]:
        h = torch.randn(B, T, d, generator=gen)
        mask = (torch.rand(B, T, generator=gen) > 0.4).to(torch.int64)
        if B > 1:
            mask[1] = 0
        href = h.clone().requires_grad_(True)
        want = (href * mask.unsqueeze(-1)).sum(1) / mask.sum(1, keepdim=True).clamp(min=1)
        gout = torch.randn(B, d, generator=gen)
        want.backward(gout)
        hd = dev(h).requires_grad_(True)
        got = ops.masked_mean(hd, dev(mask))
        got.backward(dev(gout))
        close(got, want.detach(), 1e-5, 1e-6)
        close(hd.grad, href.grad, 1e-6, 1e-7)


# ------------------------------------------------------------------------------------------
# FILIP
# ------------------------------------------------------------------------------------------
def _filip_fwd(K, hA, hB, mA, mB, t, normalise, unit=None):
    unit = normalise if unit is None else unit
    hnA, mwA = K.filip_prep(dev(hA), dev(mA), normalise, tsign=dev(t.reshape(1)))
    hnB, mwB = K.filip_prep(dev(hB), dev(mB), normalise)
    out = K.filip_sim_fwd(hnA, mwA, hA.shape[1], hnB, mwB, hB.shape[1], dev(t.reshape(1)), unit_norm=unit)
    return (hnA, mwA, hnB, mwB) + out


def test_filip_prep_layout(K):
    gen = torch.Generator().manual_seed(5)
    h = torch.randn(3, 37, 20, generator=gen)
    m = (torch.rand(3, 37, generator=gen) > 0.3).to(torch.uint8)
    hn, mw = K.filip_prep(dev(h), dev(m), True)
    assert hn.shape == (3, 64, 128) and mw.shape == (3, 2)
    got = hn.cpu().view(torch.bfloat16).float()
    want = torch.zeros(3, 64, 128)
    want[:, :37, :20] = O.bf16_round(O.filip_normalise(h)) * m[..., None]
    close(got, want, 0, 0)
    bits = mw.cpu().to(torch.int64) & 0xFFFFFFFF
    for b in range(3):
        for tok in range(64):
            on = tok < 37 and bool(m[b, tok])
            assert ((int(bits[b, tok // 32]) >> (tok % 32)) & 1) == int(on)


def test_filip_scores_match_golden(K, filip_cases):
    for key, c in filip_cases.items():
        pre = key.endswith("bf16n")
        res = _filip_fwd(K, c["hA"], c["hB"], c["maskA"], c["maskB"], c["t"], normalise=not pre)
        simA, simB = res[4], res[5]
        tol = 2e-6 if pre else 2e-3   # bf16-exact operands vs operands rounded to bf16 after normalising
        close(simA, c["sim_A"], tol, tol)
        close(simB, c["sim_B"], tol, tol)


def test_filip_argmax_points_at_the_maximum(K, filip_cases):
    c = filip_cases["s3_B8_TA48_TB40_d128_bf16n"]
    hA, hB, mA, mB, t = c["hA"], c["hB"], c["maskA"].bool(), c["maskB"].bool(), c["t"]
    res = _filip_fwd(K, hA, hB, c["maskA"], c["maskB"], t, normalise=False)
    argA, argB = res[6].cpu().long(), res[7].cpu().long()
    sim = torch.einsum("iad,jbd->ijab", hA, hB)
    sim = sim.masked_fill(~(mA[:, None, :, None] & mB[None, :, None, :]), float("-inf"))
    rA = sim.amax(3)
    pickA = torch.gather(sim, 3, argA[:, :, :48, None]).squeeze(3)
    ok = mA[:, None, :].expand_as(rA)
    assert torch.allclose(pickA[ok], rA[ok], rtol=0, atol=2e-6)
    rB = sim.amax(2)
    pickB = torch.gather(sim, 2, argB[:, :, None, :40]).squeeze(2)
    okb = mB[None, :, :].expand_as(rB)
    assert torch.allclose(pickB[okb], rB[okb], rtol=0, atol=2e-6)


def _oracle_filip_grads(hA, hB, mA, mB, t, normalise, gA, gB):
    """Oracle gradients of sum(gA*sim_A + gB*sim_B) with the operands rounded to bf16 after the
    optional normalisation (straight-through), i.e. what the matrix cores are fed."""
    a = hA.clone().requires_grad_(True)
    b = hB.clone().requires_grad_(True)
    tt = t.clone().requires_grad_(True)
    na = O.filip_normalise(a) if normalise else a
    nb = O.filip_normalise(b) if normalise else b
    na = na + (O.bf16_round(na.detach()) - na.detach())
    nb = nb + (O.bf16_round(nb.detach()) - nb.detach())
    sA, sB = O.filip_similarity_block(na, nb, mA, mB, tt)
    ((sA * gA).sum() + (sB * gB).sum()).backward()
    # the reference gives masked tokens a zero gradient; so does the kernel
    return sA.detach(), sB.detach(), a.grad, b.grad, tt.grad


def test_filip_backward_matches_oracle(K, filip_cases):
    gen = torch.Generator().manual_seed(21)
    for key, c in filip_cases.items():
        pre = key.endswith("bf16n")
        hA, hB, mA, mB, t = c["hA"], c["hB"], c["maskA"], c["maskB"], c["t"]
        B = hA.shape[0]
        gA = torch.randn(B, B, generator=gen)
        gB = torch.randn(B, B, generator=gen)
        sA, sB, dA_ref, dB_ref, dt_ref = _oracle_filip_grads(hA, hB, mA, mB, t, not pre, gA, gB)
        hnA, mwA, hnB, mwB, simA, simB, argA, argB = _filip_fwd(K, hA, hB, mA, mB, t, normalise=not pre)
        close(simA, sA, 2e-6, 2e-6)
        close(simB, sB, 2e-6, 2e-6)
        dnA, dnB, dt = K.filip_sim_bwd(hnA, mwA, hA.shape[1], hnB, mwB, hB.shape[1], dev(t.reshape(1)), simA, simB, argA, argB,
                                       dev(gA), dev(gB))
        dA = K.filip_prep_bwd(dev(hA), dnA, normalise=not pre, tsign=dev(t.reshape(1)))
        dB = K.filip_prep_bwd(dev(hB), dnB, normalise=not pre)
        # measured on MI355X (tools/measure_bwd_outliers.py): no element outside, relative L2 error 1e-7..2e-7
        mostly_close(dA, dA_ref, 1e-3, 1e-6, max_bad_frac=0.0, max_rel_l2=1e-5)
        mostly_close(dB, dB_ref, 1e-3, 1e-6, max_bad_frac=0.0, max_rel_l2=1e-5)
        close(dt, dt_ref.reshape(1), 1e-4, 1e-5)


def test_filip_backward_partner_windows(K):
    """The backward cuts the partner sequences into windows of 4 MB of images (four per gather
    launch, later launches accumulating) as soon as one side's images exceed that.  70 sequences of
    up to 256 tokens are 4.6 MB a side: two windows with partners, two empty ones, a batch size that
    is no multiple of four, one empty sequence.  Routing and arithmetic as in every other case
    (a row sums 70 to a few hundred signed terms in f32: 4e-6 of the result's norm, measured 2.6e-6)."""
    gen = torch.Generator().manual_seed(5)
    B, TA, TB, d = 70, 250, 256, 128
    hA, hB = torch.randn(B, TA, d, generator=gen), torch.randn(B, TB, d, generator=gen)
    lA, lB = torch.randint(1, TA + 1, (B,), generator=gen), torch.randint(1, TB + 1, (B,), generator=gen)
    lB[3] = 0
    mA = (torch.arange(TA)[None] < lA[:, None]).to(torch.uint8)
    mB = (torch.arange(TB)[None] < lB[:, None]).to(torch.uint8)
    gA, gB = torch.randn(B, B, generator=gen), torch.randn(B, B, generator=gen)
    gA[:, 3] = 0   # scores against the empty sequence are -inf / 0: keep them out of dt
    gB[:, 3] = 0
    res = _check_routing_and_backward(K, hA, hB, mA, mB, torch.tensor(1.0), True, gA, gB, "windows", rtol=4e-6)
    assert torch.isfinite(res[8]).all() and torch.isfinite(res[9]).all() and res[8].abs().sum() > 0


def test_filip_backward_one_side_windowed(K):
    """8 A sequences against 70 B sequences: the A side's partners (4.6 MB of B images) are cut into windows,
    the B side's (0.5 MB of A images) are not, and both sides' gathers still share one launch."""
    gen = torch.Generator().manual_seed(6)
    Br, Bc, TA, TB, d = 8, 70, 256, 256, 128
    hA, hB = torch.randn(Br, TA, d, generator=gen), torch.randn(Bc, TB, d, generator=gen)
    lA, lB = torch.randint(1, TA + 1, (Br,), generator=gen), torch.randint(1, TB + 1, (Bc,), generator=gen)
    mA = (torch.arange(TA)[None] < lA[:, None]).to(torch.uint8)
    mB = (torch.arange(TB)[None] < lB[:, None]).to(torch.uint8)
    gA, gB = torch.randn(Br, Bc, generator=gen), torch.randn(Br, Bc, generator=gen)
    res = _check_routing_and_backward(K, hA, hB, mA, mB, torch.tensor(0.7), True, gA, gB, "one side windowed", rtol=4e-6)
    assert torch.isfinite(res[8]).all() and torch.isfinite(res[9]).all()


def _check_routing_and_backward(K, hA, hB, mA, mB, t, normalise, gA, gB, tag="", rtol=2e-6):
    """The two halves of backward parity, each tight: (1) every valid token's chosen partner attains
    the maximum of its row / column of the similarity tile to 2e-6 (the forward's tolerance);
    (2) with the routing fixed the gradients equal the routed weighted sums of operand rows to f32
    summation accuracy.  Returns what the kernels produced."""
    TA, TB = hA.shape[1], hB.shape[1]
    hnA, mwA, hnB, mwB, simA, simB, argA, argB = _filip_fwd(K, hA, hB, mA, mB, t, normalise=normalise)
    nA, nB = decode_image(hnA, TA), decode_image(hnB, TB)   # the A image is negated when t < 0
    mAd, mBd = dev(mA).bool(), dev(mB).bool()
    sim = torch.einsum("iad,jbd->ijab", nA, nB).masked_fill(~(mAd[:, None, :, None] & mBd[None, :, None, :]), float("-inf"))
    liveA, liveB = mAd.any(1), mBd.any(1)
    okA = mAd[:, None, :] & liveB[None, :, None]
    okB = mBd[None, :, :] & liveA[:, None, None]
    pickA = torch.gather(sim, 3, argA[:, :, :TA, None].long().clamp(0, TB - 1)).squeeze(3)
    pickB = torch.gather(sim, 2, argB[:, :, None, :TB].long().clamp(0, TA - 1)).squeeze(2)
    assert ((sim.amax(3) - pickA)[okA].abs() <= 2e-6).all(), tag
    assert ((sim.amax(2) - pickB)[okB].abs() <= 2e-6).all(), tag
    dnA, dnB, dt = K.filip_sim_bwd(hnA, mwA, TA, hnB, mwB, TB, dev(t.reshape(1)), simA, simB, argA, argB, dev(gA), dev(gB))
    wantA, wantB = routed_grads(nA, nB, mAd, mBd, argA, argB, dev(gA), dev(gB), abs(float(t)))
    for got, want in ((dnA[:, :TA], wantA), (dnB[:, :TB], wantB)):
        err = (got.double() - want).norm(dim=-1)
        assert (err <= rtol * want.norm(dim=-1) + 1e-10 * want.abs().max()).all(), tag
        assert (got[~(want.abs().sum(-1) > 0)] == 0).all(), tag
    return hnA, mwA, hnB, mwB, simA, simB, argA, argB, dnA, dnB, dt


def test_filip_backward_is_exact_given_the_routing(K, filip_cases):
    """Every golden case: routing valid to 2e-6, arithmetic exact to f32 summation (see the helper)."""
    gen = torch.Generator().manual_seed(44)
    for key, c in filip_cases.items():
        B = c["hA"].shape[0]
        gA, gB = torch.randn(B, B, generator=gen), torch.randn(B, B, generator=gen)
        _check_routing_and_backward(K, c["hA"], c["hB"], c["maskA"], c["maskB"], c["t"], not key.endswith("bf16n"), gA, gB, key)


@pytest.mark.parametrize("unit", [False, True])
def test_filip_exact_ties_sum_to_the_reference_split(K, unit):
    """Duplicated tokens on both sides (fixture from the reference, oracle/gen_golden.py): its
    ``amax`` backward splits a gradient evenly among tied maxima (models.py:342-343), the kernels
    route it to one of the tied tokens.  Documented behaviour, pinned here: scores identical; the
    gradient summed over each group of identical tokens equals the reference's (so does every
    token outside the groups, and the whole other side, whose gradient row is the same whichever
    twin was picked).  Both score kernels: exact-f32 maxima and the integer-key unit-norm one."""
    c = load_flat("filip_ties.npz")
    hA, hB, mA, mB, t = c["hA"], c["hB"], c["maskA"], c["maskB"], c["t"]
    TA, TB = hA.shape[1], hB.shape[1]
    a = hA.clone().requires_grad_(True)
    b = hB.clone().requires_grad_(True)
    sA, sB = O.filip_similarity_block(a, b, mA, mB, t)
    gA, gB = torch.autograd.grad(O.filip_loss(sA, sB), (sA, sB), retain_graph=True)
    hnA, mwA, hnB, mwB, simA, simB, argA, argB = _filip_fwd(K, hA, hB, mA, mB, t, normalise=False, unit=unit)
    close(simA, c["sim_A"], 2e-6, 2e-6)
    close(simB, c["sim_B"], 2e-6, 2e-6)
    dnA, dnB, dt = K.filip_sim_bwd(hnA, mwA, TA, hnB, mwB, TB, dev(t.reshape(1)), simA, simB, argA, argB, dev(gA), dev(gB))
    close(dt, c["d_t"].reshape(1), 1e-4, 1e-6)

    def grouped(x, group):   # sum the rows of every group of identical tokens into its representative
        out = torch.zeros_like(x)
        out.index_add_(1, group.long(), x)
        return out

    for got, want, group in ((dnA[:, :TA, :].cpu(), c["d_hA"], c["groupA"]), (dnB[:, :TB, :].cpu(), c["d_hB"], c["groupB"])):
        g_got, g_want = grouped(got, group), grouped(want, group)
        mostly_close(g_got, g_want, 1e-4, 1e-8, max_bad_frac=0.002, max_rel_l2=2e-3)
    # and the split itself differs: some twin pair does not carry equal shares in the kernel's result
    twins = [(int(tw), int(c["groupB"][tw])) for tw in (c["groupB"] != torch.arange(TB)).nonzero().flatten()]
    assert any(not torch.equal(dnB[:, tw, :].cpu(), dnB[:, rep, :].cpu()) for tw, rep in twins)


def test_filip_negative_temperature(K, filip_cases):
    c = filip_cases["s1_B6_TA9_TB11_d16_raw"]
    t = torch.tensor(-0.9)
    sA, sB = O.filip_similarity(O.bf16_round(O.filip_normalise(c["hA"])), O.bf16_round(O.filip_normalise(c["hB"])),
                                c["maskA"], c["maskB"], t)
    res = _filip_fwd(K, c["hA"], c["hB"], c["maskA"], c["maskB"], t, normalise=True)
    close(res[4], sA, 2e-6, 2e-6)
    close(res[5], sB, 2e-6, 2e-6)


def test_filip_all_padding_sequence(K):
    """A sequence with no valid token: the reference fills its pairs with finfo.min; sums of
    those overflow to -inf on the side that averages over the OTHER sequence's tokens, and the
    empty side averages to zero."""
    gen = torch.Generator().manual_seed(9)
    hA = torch.randn(3, 6, 16, generator=gen); hB = torch.randn(3, 5, 16, generator=gen)
    mA = torch.ones(3, 6, dtype=torch.uint8); mB = torch.ones(3, 5, dtype=torch.uint8)
    mB[1] = 0
    t = torch.tensor(1.0)
    sA, sB = O.filip_similarity(O.bf16_round(O.filip_normalise(hA)), O.bf16_round(O.filip_normalise(hB)), mA, mB, t)
    res = _filip_fwd(K, hA, hB, mA, mB, t, normalise=True)
    gotA, gotB = res[4].cpu(), res[5].cpu()
    assert torch.isinf(sA[:, 1]).all() and torch.isinf(gotA[:, 1]).all() and (gotA[:, 1] < 0).all()
    assert (sB[:, 1] == 0).all() and (gotB[:, 1] == 0).all()
    keep = [0, 2]
    close(gotA[:, keep], sA[:, keep], 2e-6, 2e-6)
    close(gotB[:, keep], sB[:, keep], 2e-6, 2e-6)


def test_filip_benchmark_size_matches_reference_scores(K):
    """B = 256, L = 256, d = 128: every one of the 2 x 65,536 scores against the reference's
    (generated block-wise by oracle/gen_golden.py), plus the loss through the row-CE kernels."""
    g = load_flat("filip_b256_l256.npz")
    B, L, d = 256, 256, 128
    gen = torch.Generator().manual_seed(0)
    hA = O.bf16_round(torch.randn(B, L, d, generator=gen))
    hB = O.bf16_round(torch.randn(B, L, d, generator=gen))
    lA = torch.randint(128, 257, (B,), generator=gen)
    lB = torch.randint(128, 257, (B,), generator=gen)
    mA, mB = O.right_padded_masks(lA, L), O.right_padded_masks(lB, L)
    res = _filip_fwd(K, hA, hB, mA, mB, torch.tensor(1.0), normalise=True)
    simA, simB = res[4], res[5]
    close(simA, g["sim_A"], 2e-3, 2e-3)
    close(simB, g["sim_B"], 2e-3, 2e-3)
    _, lossA = K.rowce_fwd(simA)
    _, lossB = K.rowce_fwd(simB)
    close((lossA + lossB) / 2, g["loss"].reshape(1), 1e-3, 1e-3)
    # The fixture above comes from f32-normalised tokens, which the kernel rounds to bf16 (hence 2e-3).  The second one was
    # generated from the rounded unit vectors themselves (normalise in f32, round to bf16, then the reference's
    # _filip_similarity_score): identical operands on both sides, so ALL 2 x 65,536 scores and the loss are compared at the
    # tolerance of the small cases.
    gn = load_flat("filip_b256_l256_bf16n.npz")
    # the generator's own expression (torch's norm; the oracle's sqrt(sum(h^2)) differs from it by an ulp now and then, which
    # flips the bf16 rounding of about one element in 1e5 and moves a score by 2e-6)
    nA = (hA / hA.norm(dim=-1, keepdim=True)).to(torch.bfloat16).float()
    nB = (hB / hB.norm(dim=-1, keepdim=True)).to(torch.bfloat16).float()
    res_n = _filip_fwd(K, nA, nB, mA, mB, torch.tensor(1.0), normalise=False, unit=True)
    # measured over all 131,072 scores (tools/score_error_b256.py, profiles/r3_score_error_b256.txt): median 3e-8, largest 1.94e-7
    close(res_n[4], gn["sim_A"], 1e-6, 1e-6)
    close(res_n[5], gn["sim_B"], 1e-6, 1e-6)
    _, lossAn = K.rowce_fwd(res_n[4])
    _, lossBn = K.rowce_fwd(res_n[5])
    close((lossAn + lossBn) / 2, gn["loss"].reshape(1), 1e-5, 1e-5)
    # normalising on the device (f32 normalise, bf16 round inside pclip_filip_prep) gives the same scores up to such flips
    close(res_n[4], simA.cpu(), 5e-6, 5e-6)
    close(res_n[5], simB.cpu(), 5e-6, 5e-6)
    # size-independent properties: transposing the roles of the two sides transposes the scores
    res_t = _filip_fwd(K, hB, hA, mB, mA, torch.tensor(1.0), normalise=True)
    assert torch.equal(res_t[5].t().contiguous(), simA) or torch.allclose(res_t[5].t(), simA, atol=1e-6)
    assert torch.allclose(res_t[4].t(), simB, atol=1e-6)
    # and the scores of a sub-batch are the corresponding block of the full result
    sub = _filip_fwd(K, hA[:16], hB[32:48], mA[:16], mB[32:48], torch.tensor(1.0), normalise=True)
    assert torch.allclose(sub[4], simA[:16, 32:48], atol=1e-6) and torch.allclose(sub[5], simB[:16, 32:48], atol=1e-6)


def test_filip_benchmark_size_backward_rows_against_oracle(K):
    """B = 256, L = 256, d = 128 (eight partner windows in the backward): the gradients of two
    whole A sequences and two whole B sequences against the oracle's autograd of exactly the
    pairs they take part in, plus two size-independent properties over every token: the gradient
    of a normalised token is orthogonal to it, and padding tokens get none.  The arithmetic of the
    backward is exact once the routing is fixed (checked to 2e-6 per row here as well); what this
    comparison adds is the routing itself against the oracle's own arg-max.  A token's gradient row
    sums about 500 routed contributions, and where two candidate partners lie closer than the f32
    rounding of the two different summation orders (kernel: accumulation from 4.0 on the matrix
    cores; oracle: an f32 einsum) the two sides may route one contribution differently, which moves
    that whole 128-wide row a little: a few percent of the elements sit outside the elementwise
    tolerance while the relative L2 error stays small."""
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
    for i in (3, 200):   # A sequence i against every B sequence
        sA, sB, da, _, _ = _oracle_filip_grads(hA[i:i + 1], hB, mA[i:i + 1], mB, tt, True, gA[i:i + 1], gB[i:i + 1])
        close(simA[i:i + 1], sA, 3e-5, 3e-5)   # 1/t = 14 scales the 2e-6 of the unit-temperature scores
        close(simB[i:i + 1], sB, 3e-5, 3e-5)
        # budgets = twice what tools/measure_bwd_outliers.py measured (profiles/r3_bwd_outliers.txt: <= 0.03 %, 1e-4; every
        # outlier row is a token routed differently from the oracle at a gap below 1.5e-6, or one whose bf16 operand the
        # device and the host rounded differently)
        mostly_close(dA[i:i + 1], da, 1e-3, 1e-7, max_bad_frac=0.0007, max_rel_l2=2e-4)
    for j in (0, 131):   # every A sequence against B sequence j
        _, _, _, db, _ = _oracle_filip_grads(hA, hB[j:j + 1], mA, mB[j:j + 1], tt, True, gA[:, j:j + 1], gB[:, j:j + 1])
        mostly_close(dB[j:j + 1], db, 1e-3, 1e-7, max_bad_frac=0.023, max_rel_l2=0.016)   # measured: <= 1.14 %, 7.95e-3
    for h, dh, m in ((hA, dA, mA), (hB, dB, mB)):
        assert (dh[~m.bool()] == 0).all()
        dots = (dh * h).sum(-1)[m.bool()]
        scale = (dh.norm(dim=-1) * h.norm(dim=-1))[m.bool()]
        assert (dots.abs() <= 1e-4 * scale + 1e-9).all()


def test_filip_rank_shape_decomposes_over_partner_blocks(K):
    """One rank of a 3-GPU job: 64 A sequences against 3 x 256 gathered B sequences, L = 256
    (the backward then needs three gather launches of four 4-MB windows each).  The two
    directions are independent per sequence pair (models.py:332-351), so the scores of a column
    block equal the scores computed against that block alone, and the A gradient is the sum of
    the per-block A gradients."""
    gen = torch.Generator().manual_seed(11)
    Br, Bc, L, d, blk = 64, 768, 256, 128, 256
    hA = O.bf16_round(torch.randn(Br, L, d, generator=gen))
    hB = O.bf16_round(torch.randn(Bc, L, d, generator=gen))
    mA = O.right_padded_masks(torch.randint(100, L + 1, (Br,), generator=gen), L)
    mB = O.right_padded_masks(torch.randint(100, L + 1, (Bc,), generator=gen), L)
    t = dev(torch.tensor([1.0]))
    gA, gB = dev(torch.randn(Br, Bc, generator=gen) / Bc), dev(torch.randn(Br, Bc, generator=gen) / Bc)
    hnA, mwA = K.filip_prep(dev(hA), dev(mA), True, tsign=t)
    hnB, mwB = K.filip_prep(dev(hB), dev(mB), True)
    simA, simB, argA, argB = K.filip_sim_fwd(hnA, mwA, L, hnB, mwB, L, t, unit_norm=True)
    dnA, dnB, dt = K.filip_sim_bwd(hnA, mwA, L, hnB, mwB, L, t, simA, simB, argA, argB, gA, gB)
    dnA_sum = torch.zeros_like(dnA)
    dt_sum = torch.zeros_like(dt)
    for k in range(Bc // blk):
        sl = slice(k * blk, (k + 1) * blk)
        sa, sb, aa, ab = K.filip_sim_fwd(hnA, mwA, L, hnB[sl].contiguous(), mwB[sl].contiguous(), L, t, unit_norm=True)
        assert torch.equal(sa, simA[:, sl]) and torch.equal(sb, simB[:, sl])
        da, db, dtk = K.filip_sim_bwd(hnA, mwA, L, hnB[sl].contiguous(), mwB[sl].contiguous(), L, t, sa, sb, aa, ab,
                                      gA[:, sl].contiguous(), gB[:, sl].contiguous())
        dnA_sum += da
        dt_sum += dtk
        close(dnB[sl], db.cpu(), 1e-5, 1e-7)
    close(dnA, dnA_sum.cpu(), 1e-4, 1e-7)
    close(dt, dt_sum.cpu(), 1e-4, 1e-6)


@pytest.mark.parametrize("TA,TB,unit", [(256, 200, True), (70, 96, False), (300, 280, True)])
def test_filip_scores_in_column_blocks_fill_the_wide_outputs(K, TA, TB, unit):
    """pclip_filip_sim_fwd_cols: three calls on column blocks of uneven width (the middle one first, as a rank scores
    its own sequences before the gathered ones) leave in the wide outputs exactly what one call on all columns
    returns — scores and arg indices, for the streaming, the generic and the long-sequence kernel."""
    gen = torch.Generator().manual_seed(31)
    Br, Bc, d = 10, 23, 128
    hA, hB = O.bf16_round(torch.randn(Br, TA, d, generator=gen)), O.bf16_round(torch.randn(Bc, TB, d, generator=gen))
    mA = O.right_padded_masks(torch.randint(1, TA + 1, (Br,), generator=gen), TA)
    mB = O.right_padded_masks(torch.randint(1, TB + 1, (Bc,), generator=gen), TB)
    t = dev(torch.tensor([0.8]))
    hnA, mwA = K.filip_prep(dev(hA), dev(mA), True, tsign=t)
    hnB, mwB = K.filip_prep(dev(hB), dev(mB), True)
    want = K.filip_sim_fwd(hnA, mwA, TA, hnB, mwB, TB, t, unit_norm=unit)
    TAp, TBp = (TA + 31) // 32 * 32, (TB + 31) // 32 * 32
    out = (torch.full((Br, Bc), float("nan"), device="cuda"), torch.full((Br, Bc), float("nan"), device="cuda"),
           torch.full((Br, Bc, TAp), -1, dtype=torch.int16, device="cuda"), torch.full((Br, Bc, TBp), -1, dtype=torch.int16, device="cuda"))
    for lo, hi in ((9, 16), (0, 9), (16, 23)):
        K.filip_sim_fwd(hnA, mwA, TA, hnB[lo:hi], mwB[lo:hi], TB, t, unit_norm=unit, out=out, col_offset=lo)
    assert torch.equal(out[0], want[0]) and torch.equal(out[1], want[1])
    vA = dev(mA).bool()[:, None, :] & (dev(mB).sum(1) > 0)[None, :, None]
    vB = dev(mB).bool()[None, :, :] & (dev(mA).sum(1) > 0)[:, None, None]
    assert torch.equal(out[2][..., :TA][vA], want[2][..., :TA][vA]) and torch.equal(out[3][..., :TB][vB], want[3][..., :TB][vB])
    with pytest.raises(Exception):
        K.filip_sim_fwd(hnA, mwA, TA, hnB[:8], mwB[:8], TB, t, unit_norm=unit, out=out, col_offset=16)   # columns 16 .. 23 of 23


def test_filip_long_segments_agree_with_the_general_kernel(K):
    """Sequences beyond 256 tokens, unit norm: the segmented run on the streaming kernel (pclip_filip_sim_fwd_ws with a
    workspace) against the general long-sequence kernel (the same entry without one) — scores to 2e-6, and every arg
    index of either run attains the row / column maximum of the float64 similarities to 2e-6.  Ragged lengths, one
    sequence shorter than a segment, one empty, a column block of a wider output."""
    from protein_clip_amd import _lib
    from protein_clip_amd._lib import ptr, stream_of
    lib = _lib.load()
    gen = torch.Generator().manual_seed(77)
    Br, Bc, TA, TB, d = 5, 6, 1000, 700, 128
    hA, hB = O.bf16_round(torch.randn(Br, TA, d, generator=gen)), O.bf16_round(torch.randn(Bc, TB, d, generator=gen))
    lA = torch.tensor([1000, 513, 100, 777, 256])
    lB = torch.tensor([700, 0, 257, 31, 512, 699])
    mA, mB = O.right_padded_masks(lA, TA), O.right_padded_masks(lB, TB)
    t = dev(torch.tensor([1.0]))
    hnA, mwA = K.filip_prep(dev(hA), dev(mA), True, tsign=t)
    hnB, mwB = K.filip_prep(dev(hB), dev(mB), True)
    TAp, TBp, wide = 1024, 704, 9
    nws = lib.pclip_filip_sim_fwd_workspace_bytes(Br, TA, Bc, TB, 1)
    assert nws > 0 and lib.pclip_filip_sim_fwd_workspace_bytes(Br, 200, Bc, 256, 1) == 0
    res = []
    for ws in (torch.empty(nws, dtype=torch.uint8, device="cuda"), None):
        sA, sB = torch.full((Br, wide), float("nan"), device="cuda"), torch.full((Br, wide), float("nan"), device="cuda")
        aA = torch.full((Br, wide, TAp), -1, dtype=torch.int16, device="cuda")
        aB = torch.full((Br, wide, TBp), -1, dtype=torch.int16, device="cuda")
        rc = lib.pclip_filip_sim_fwd_ws(ptr(hnA), ptr(mwA), Br, TA, ptr(hnB), ptr(mwB), Bc, TB, ptr(t), 1, ptr(sA), ptr(sB), ptr(aA), ptr(aB), 2, wide,
                                        ptr(ws), nws if ws is not None else 0, stream_of(hnA))
        assert rc == 0, lib.pclip_last_error()
        torch.cuda.synchronize()
        assert torch.isnan(sA[:, :2]).all() and torch.isnan(sA[:, 8:]).all()   # only the block's columns are written
        res.append((sA[:, 2:8], sB[:, 2:8], aA[:, 2:8], aB[:, 2:8]))
    nA = hnA.view(torch.bfloat16).double()[:, :TA]
    nB = hnB.view(torch.bfloat16).double()[:, :TB]
    S = torch.einsum("iad,jbd->ijab", nA, nB)
    ok = dev(mA).bool()[:, None, :, None] & dev(mB).bool()[None, :, None, :]
    S = S.masked_fill(~ok, -1e30)
    for sA, sB, aA, aB in res:
        rows = dev(mA).bool()[:, None, :] & (dev(lB) > 0)[None, :, None]
        cols = dev(mB).bool()[None, :, :] & (dev(lA) > 0)[:, None, None]
        gotr = torch.gather(S, 3, aA[..., :TA].long().clamp(0, TB - 1)[..., None])[..., 0]
        gotc = torch.gather(S, 2, aB[..., :TB].long().clamp(0, TA - 1)[:, :, None, :])[:, :, 0, :]
        assert ((S.amax(3) - gotr)[rows].abs() <= 2e-6).all() and ((S.amax(2) - gotc)[cols].abs() <= 2e-6).all()
        wantA = torch.where(rows, S.amax(3), torch.zeros((), dtype=torch.float64, device="cuda")).sum(-1) / dev(lA).clamp(min=1)[:, None]
        wantB = torch.where(cols, S.amax(2), torch.zeros((), dtype=torch.float64, device="cuda")).sum(-1) / dev(lB).clamp(min=1)[None, :]
        keep = dev(lB) > 0   # an empty B sequence: minus infinity on the row side
        close(sA[:, keep], wantA[:, keep].float().cpu(), 2e-6, 2e-6)
        close(sB[:, keep], wantB[:, keep].float().cpu(), 2e-6, 2e-6)
        assert torch.isinf(sA[:, ~keep]).all() and (sB[:, ~keep] == 0).all()
    close(res[0][0][:, keep], res[1][0][:, keep].cpu(), 2e-6, 2e-6)
    close(res[0][1], res[1][1].cpu(), 2e-6, 2e-6)


def test_filip_long_segments_in_workspace_bounded_column_blocks(K):
    """A long-sequence problem whose segment tables exceed the library's workspace budget (32 x 768 sequences of about
    1,000 tokens: the launch runs in column blocks inside the one call) against the same problem scored in explicit
    column blocks that each stay under the budget, and against a call without arg tables: identical results."""
    from protein_clip_amd import _lib
    gen = torch.Generator().manual_seed(5)
    Br, Bc, TA, TB, d = 32, 768, 1000, 1024, 128
    hA, hB = dev(torch.randn(Br, TA, d, generator=gen)), dev(torch.randn(Bc, TB, d, generator=gen))
    lA, lB = torch.randint(300, TA + 1, (Br,), generator=gen), torch.randint(1, TB + 1, (Bc,), generator=gen)
    lB[3] = 0
    mA, mB = dev(O.right_padded_masks(lA, TA)), dev(O.right_padded_masks(lB, TB))
    t = dev(torch.tensor([0.9]))
    hnA, mwA = K.filip_prep(hA, mA, True, tsign=t)
    hnB, mwB = K.filip_prep(hB, mB, True)
    lib = _lib.load()
    whole = lib.pclip_filip_sim_fwd_workspace_bytes(Br, TA, Bc, TB, 1)
    assert whole < 6 * Br * Bc * (1024 + 1024)   # bounded: less than the tables of all columns at once
    sA, sB, aA, aB = K.filip_sim_fwd(hnA, mwA, TA, hnB, mwB, TB, t, want_arg=True, unit_norm=True)
    out = (torch.empty_like(sA), torch.empty_like(sB), torch.empty_like(aA), torch.empty_like(aB))
    for c0 in range(0, Bc, 96):
        c1 = min(Bc, c0 + 96)
        assert lib.pclip_filip_sim_fwd_workspace_bytes(Br, TA, c1 - c0, TB, 1) < whole
        K.filip_sim_fwd(hnA, mwA, TA, hnB[c0:c1], mwB[c0:c1], TB, t, want_arg=True, unit_norm=True, out=out, col_offset=c0)
    keep = dev(lB > 0)
    assert torch.equal(sA[:, keep], out[0][:, keep]) and torch.equal(sB, out[1])
    assert torch.isinf(sA[:, 3]).all() and (sB[:, 3] == 0).all()
    vA = mA.bool()[:, None, :] & keep[None, :, None]
    vB = mB.bool()[None, :, :].expand(Br, Bc, TB)
    assert not ((aA[..., :TA] != out[2][..., :TA]) & vA).any() and not ((aB[..., :TB] != out[3][..., :TB]) & vB).any()
    s2A, s2B, _, _ = K.filip_sim_fwd(hnA, mwA, TA, hnB, mwB, TB, t, want_arg=False, unit_norm=True)
    assert torch.equal(sA[:, keep], s2A[:, keep]) and torch.equal(sB, s2B)


@pytest.mark.parametrize("TA,TB,normalise", [(300, 270, True), (40, 333, True), (290, 64, False), (1000, 1000, True), (2000, 600, True)])
def test_filip_sequences_beyond_one_lds_image(K, TA, TB, normalise):
    """Chains longer than 256 tokens (the reference keeps up to 2,000 residues,
    data_utils_2protein.py:88): scores, arg-max routing and the backward against the oracle, ragged masks,
    one empty middle stretch.  Normalised operands run as 256-token segments on the streaming kernel,
    others on the general kernel."""
    gen = torch.Generator().manual_seed(TA * 1000 + TB)
    B, d = 3, 128
    hA, hB = torch.randn(B, TA, d, generator=gen), torch.randn(B, TB, d, generator=gen)
    mA = O.right_padded_masks(torch.randint(TA // 2, TA + 1, (B,), generator=gen), TA).clone()
    mB = O.right_padded_masks(torch.randint(TB // 2, TB + 1, (B,), generator=gen), TB).clone()
    mA[1, 30:70] = 0   # a whole tile of padding inside a sequence
    t = torch.tensor(0.8)
    gA, gB = torch.randn(B, B, generator=gen), torch.randn(B, B, generator=gen)
    if not normalise:   # pre-normalised, bf16-exact operands through the raw entry
        hA, hB = O.bf16_round(O.filip_normalise(hA)), O.bf16_round(O.filip_normalise(hB))
    sA, sB, dA_ref, dB_ref, dt_ref = _oracle_filip_grads(hA, hB, mA, mB, t, normalise, gA, gB)
    hnA, mwA, hnB, mwB, simA, simB, argA, argB = _filip_fwd(K, hA, hB, mA, mB, t, normalise=normalise)
    close(simA, sA, 2e-6, 2e-6)
    close(simB, sB, 2e-6, 2e-6)
    dnA, dnB, dt = K.filip_sim_bwd(hnA, mwA, TA, hnB, mwB, TB, dev(t.reshape(1)), simA, simB, argA, argB, dev(gA), dev(gB))
    dA = K.filip_prep_bwd(dev(hA), dnA, normalise=normalise, tsign=dev(t.reshape(1)))
    dB = K.filip_prep_bwd(dev(hB), dnB, normalise=normalise)
    mostly_close(dA, dA_ref, 1e-3, 1e-6, max_bad_frac=0.005, max_rel_l2=0.01)
    mostly_close(dB, dB_ref, 1e-3, 1e-6, max_bad_frac=0.005, max_rel_l2=0.01)
    close(dt, dt_ref.reshape(1), 1e-4, 1e-5)


def test_filip_backward_in_two_one_sided_calls(K):
    """The multi-GPU step's split (steps.py): B side first, then the A side resumed on the same workspace
    (pclip_filip_sim_bwd_resume keeps the first call's marks of mutual choices): both equal the one call for both
    sides, single window and windowed."""
    for (Br, Bc, T) in [(5, 7, 40), (40, 72, 256)]:
        gen = torch.Generator().manual_seed(Br)
        hA, hB = torch.randn(Br, T, 128, generator=gen), torch.randn(Bc, T, 128, generator=gen)
        lA, lB = torch.randint(1, T + 1, (Br,), generator=gen), torch.randint(1, T + 1, (Bc,), generator=gen)
        mA = (torch.arange(T)[None] < lA[:, None]).to(torch.uint8)
        mB = (torch.arange(T)[None] < lB[:, None]).to(torch.uint8)
        t = torch.tensor(0.9)
        hnA, mwA, hnB, mwB, simA, simB, argA, argB = _filip_fwd(K, hA, hB, mA, mB, t, normalise=True)
        gA, gB = dev(torch.randn(Br, Bc, generator=gen)), dev(torch.randn(Br, Bc, generator=gen))
        tt = dev(t.reshape(1))
        dnA, dnB, dt = K.filip_sim_bwd(hnA, mwA, T, hnB, mwB, T, tt, simA, simB, argA, argB, gA, gB)
        ws = K.filip_bwd_workspace(Br, T, Bc, T, "cuda")   # the pair's own: the second call finds the first one's marks in it
        _, dnB2, _ = K.filip_sim_bwd(hnA, mwA, T, hnB, mwB, T, tt, simA, simB, argA, argB, gA, gB, sides="B", ws=ws)
        # another backward in between, on the shared scratch of the stream: the pair's workspace is not touched by it
        K.filip_sim_bwd(hnA, mwA, T, hnB, mwB, T, tt, simA, simB, argA, argB, gB, gA)
        dnA2, _, dt2 = K.filip_sim_bwd(hnA, mwA, T, hnB, mwB, T, tt, simA, simB, argA, argB, gA, gB, sides="A", resume=True, ws=ws)
        with pytest.raises(Exception):
            K.filip_sim_bwd(hnA, mwA, T, hnB, mwB, T, tt, simA, simB, argA, argB, gA, gB, sides="A", resume=True)   # no workspace to resume on
        close(dnA2, dnA.cpu(), 1e-5, 2e-6)   # the order of a bucket's entries, hence of the sums, is not fixed
        close(dnB2, dnB.cpu(), 1e-5, 2e-6)
        close(dt2, dt.cpu(), 1e-5, 1e-6)


@pytest.mark.parametrize("Br", [70, 600, 1030])
def test_filip_scores_with_many_a_sequences(K, Br):
    """Batches of many short A sequences: more pairs than one row of the score kernel's cost-ranked dealing (32),
    more than one wave of pairs (128), and beyond the size up to which the pairing runs at all (1024): ragged
    lengths with empty sequences in between, scores against the oracle."""
    gen = torch.Generator().manual_seed(Br)
    Bc, TA, TB, d = 24, 70, 45, 32
    hA, hB = torch.randn(Br, TA, d, generator=gen), torch.randn(Bc, TB, d, generator=gen)
    lA = torch.randint(0, TA + 1, (Br,), generator=gen)
    lB = torch.randint(1, TB + 1, (Bc,), generator=gen)
    lA[::7] = TA
    mA = (torch.arange(TA)[None] < lA[:, None]).to(torch.uint8)
    mB = (torch.arange(TB)[None] < lB[:, None]).to(torch.uint8)
    t = torch.tensor(0.8)
    hA, hB = O.bf16_round(O.filip_normalise(hA)), O.bf16_round(O.filip_normalise(hB))   # the same bf16 operands on both sides
    sA, sB = O.filip_similarity(hA, hB, mA, mB, t, block=64)
    hnA, mwA, hnB, mwB, simA, simB, argA, argB = _filip_fwd(K, hA, hB, mA, mB, t, normalise=False, unit=True)
    live = (lA > 0)[:, None].expand(Br, Bc)
    assert torch.allclose(simA.cpu()[live], sA[live], rtol=2e-6, atol=3e-6)
    assert torch.allclose(simB.cpu()[live], sB[live], rtol=2e-6, atol=3e-6)


def test_filip_random_small_shapes_against_oracle(K):
    """Seeded sweep over awkward shapes: batch sizes 1..19 (odd ones leave a workgroup with one A
    sequence), token counts around the 32-token tile edges, narrow embeddings, random (not only
    right-padded) masks with empty sequences on either side, both signs of the temperature,
    Br != Bc.  Scores and gradients against the oracle."""
    gen = torch.Generator().manual_seed(2024)
    pick = lambda xs: xs[int(torch.randint(len(xs), (1,), generator=gen))]
    for trial in range(24):
        Br, Bc = pick([1, 2, 3, 5, 9, 19]), pick([1, 2, 3, 5, 9, 19])
        TA, TB = pick([1, 5, 31, 32, 33, 64, 65, 100]), pick([1, 5, 31, 32, 33, 64, 65, 100])
        d = pick([8, 20, 128])
        normalise = trial % 3 != 2
        t = torch.tensor(pick([1.0, 0.7, -0.9]))
        hA, hB = torch.randn(Br, TA, d, generator=gen), torch.randn(Bc, TB, d, generator=gen)
        mA = (torch.rand(Br, TA, generator=gen) > 0.3).to(torch.uint8)
        mB = (torch.rand(Bc, TB, generator=gen) > 0.3).to(torch.uint8)
        mA[:, 0] = 1
        mB[:, 0] = 1
        if trial % 4 == 1 and Bc > 1:
            mB[Bc - 1] = 0          # an empty B sequence
        if trial % 4 == 3 and Br > 1:
            mA[0] = 0               # an empty A sequence
        if not normalise:
            hA, hB = O.bf16_round(O.filip_normalise(hA)), O.bf16_round(O.filip_normalise(hB))
        gA, gB = torch.randn(Br, Bc, generator=gen), torch.randn(Br, Bc, generator=gen)
        liveA, liveB = mA.sum(1) > 0, mB.sum(1) > 0
        gA[:, ~liveB] = 0           # scores against an empty sequence are -inf / 0: keep them out of the sums
        gB[:, ~liveB] = 0
        gA[~liveA] = 0
        gB[~liveA] = 0
        sA, sB, dA_ref, dB_ref, dt_ref = _oracle_filip_grads(hA, hB, mA, mB, t, normalise, gA, gB)
        hnA, mwA, hnB, mwB, simA, simB, argA, argB = _filip_fwd(K, hA, hB, mA, mB, t, normalise=normalise)
        ok = liveA[:, None] & liveB[None, :]
        tag = f"trial {trial}: Br={Br} Bc={Bc} TA={TA} TB={TB} d={d} normalise={normalise} t={float(t)}"
        assert torch.allclose(simA.cpu()[ok], sA[ok], rtol=2e-6, atol=3e-6), tag
        assert torch.allclose(simB.cpu()[ok], sB[ok], rtol=2e-6, atol=3e-6), tag
        dnA, dnB, dt = K.filip_sim_bwd(hnA, mwA, TA, hnB, mwB, TB, dev(t.reshape(1)), simA, simB, argA, argB, dev(gA), dev(gB))
        dA = K.filip_prep_bwd(dev(hA), dnA, normalise=normalise, tsign=dev(t.reshape(1)))
        dB = K.filip_prep_bwd(dev(hB), dnB, normalise=normalise)
        try:
            mostly_close(dA, dA_ref, 1e-3, 1e-6, max_bad_frac=0.005, max_rel_l2=0.01)
            mostly_close(dB, dB_ref, 1e-3, 1e-6, max_bad_frac=0.005, max_rel_l2=0.01)
        except AssertionError as exc:
            raise AssertionError(f"{tag}: {exc}") from None
        if bool(ok.all()):
            assert torch.allclose(dt.cpu(), dt_ref.reshape(1), rtol=1e-4, atol=1e-5), tag


@pytest.mark.parametrize("Br,Bc,TA,TB", [(11, 13, 37, 70), (40, 36, 256, 250)])
def test_filip_fused_backward_equals_the_two_calls(K, Br, Bc, TA, TB):
    """pclip_filip_bwd_fused (score backward + prep backward, partials summed inside the latter)
    against pclip_filip_sim_bwd followed by pclip_filip_prep_bwd, single window and windowed."""
    gen = torch.Generator().manual_seed(31)
    for normalise, d, tval in [(True, 128, 0.9), (False, 20, -1.1)]:
        hA, hB = torch.randn(Br, TA, d, generator=gen), torch.randn(Bc, TB, d, generator=gen)
        mA = (torch.rand(Br, TA, generator=gen) > 0.3).to(torch.uint8)
        mB = (torch.rand(Bc, TB, generator=gen) > 0.3).to(torch.uint8)
        mA[:, 0] = 1
        mB[:, 0] = 1
        t = dev(torch.tensor([tval]))
        gA, gB = dev(torch.randn(Br, Bc, generator=gen)), dev(torch.randn(Br, Bc, generator=gen))
        hAd, hBd = dev(hA), dev(hB)
        hnA, mwA = K.filip_prep(hAd, dev(mA), normalise, tsign=t)
        hnB, mwB = K.filip_prep(hBd, dev(mB), normalise)
        simA, simB, argA, argB = K.filip_sim_fwd(hnA, mwA, TA, hnB, mwB, TB, t, unit_norm=normalise)
        dnA, dnB, dt = K.filip_sim_bwd(hnA, mwA, TA, hnB, mwB, TB, t, simA, simB, argA, argB, gA, gB)
        want = (K.filip_prep_bwd(hAd, dnA, normalise, tsign=t), K.filip_prep_bwd(hBd, dnB, normalise), dt)
        got = K.filip_bwd_fused(hnA, mwA, hnB, mwB, t, simA, simB, argA, argB, gA, gB, hAd, hBd, normalise)
        for a, b in zip(got, want):
            torch.testing.assert_close(a, b, rtol=1e-5, atol=1e-7)


def test_filip_scores_do_not_depend_on_the_workgroup_count():
    """The streaming kernel cuts the B side into chunks by a workgroup target (768 by default):
    scores and arg-max tables are bit-identical for every cut."""
    import importlib.util
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("stream_wgs_check", os.path.join(root, "tools", "stream_wgs_check.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    sums = [mod.checksum(w) for w in (2, 40, 0, 5000)]
    assert len(set(sums)) == 1, sums
