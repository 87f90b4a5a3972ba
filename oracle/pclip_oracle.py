"""CPU oracle for the contrastive hot path (TEST INFRASTRUCTURE ONLY).

This file is a checker, not a product path.  Only ``tests/``, ``__graft_entry__.smoke()``
and the ``cpu_baseline`` leg of ``bench.py`` may import it; the shipped package never does
and fails loudly when its HIP library is missing.

It restates, in plain PyTorch-CPU tensor arithmetic written from the mathematics, what the
reference computes between the projection-head output and the scalar loss (and the
gradients autograd would return for it).  Each function cites the reference lines it follows
(paths relative to the reference checkout).  Parity is pinned: ``oracle/gen_golden.py``
imports the reference's own functions and stores their outputs under ``tests/golden/``;
``tests/test_oracle_golden.py`` checks every function here against those files.  The one
exception is the GradCache list boundary (``grad_cache`` is an un-vendored third-party
package, absent offline): the list-taking variants are pinned through a pass-through
stand-in for its two decorators, i.e. parity of the loss arithmetic is pinned, the
GradCache package behaviour itself is "parity unpinned".

All functions take/return ``torch.Tensor`` on CPU.  ``dtype`` of the computation follows the
inputs (float32 mirrors the reference's CPU path; float64 is used as a tighter referee).
"""
from __future__ import annotations

import math
from typing import List, Sequence, Tuple

import torch

# --------------------------------------------------------------------------------------
# CLIP tower head: pooling, embedding_ffn, normalise
# --------------------------------------------------------------------------------------


def encoder_masked_mean(h: torch.Tensor, mask: torch.Tensor) -> torch.Tensor:
    """``Encoder._masked_mean`` (modules/models.py:76-90): the mask multiplies by VALUE and the
    token count is clamped to at least 1."""
    m = mask.to(h.dtype)
    return (h * m.unsqueeze(-1)).sum(dim=1) / m.sum(dim=1, keepdim=True).clamp(min=1)


def head_tail(pooled: torch.Tensor, params: Sequence[torch.Tensor], keep, keep_scale: float, ln_eps: float,
              temperature: torch.Tensor) -> torch.Tensor:
    """``embedding_ffn`` then normalise and scale (modules/models.py:49-51; the network is built at
    :54-73: blocks of Linear -> ReLU -> LayerNorm -> Dropout, then one Linear).  ``params`` lists
    ``W, b, gamma, beta`` per block and the last ``W, b``; ``keep [blocks,B,d]`` holds the dropout
    masks (None = eval mode), a kept element is multiplied by ``keep_scale = 1 / (1 - p)``.
    LayerNorm uses the biased variance, as ``nn.LayerNorm`` does."""
    x = pooled
    for l in range((len(params) - 2) // 4):
        W, b, gamma, beta = params[4 * l:4 * l + 4]
        a = (x @ W.t() + b).clamp(min=0)
        mu = a.mean(dim=-1, keepdim=True)
        var = (a - mu).pow(2).mean(dim=-1, keepdim=True)
        x = (a - mu) / torch.sqrt(var + ln_eps) * gamma + beta
        if keep is not None:
            x = x * keep[l].to(x.dtype) * keep_scale
    return l2norm_scale(x @ params[-2].t() + params[-1], temperature)


def head_step(tokens, mask, params, keep, keep_scale, ln_eps, temperature, d_out):
    """Forward and the gradients autograd returns for ``sum(out * d_out)``:
    ``(out, pooled, d_tokens, d_pooled, [d_param ...], d_t)``."""
    tok = tokens.detach().clone().requires_grad_(True)
    prm = [p.detach().clone().requires_grad_(True) for p in params]
    t = temperature.detach().clone().requires_grad_(True)
    pooled = encoder_masked_mean(tok, mask)
    pooled.retain_grad()
    out = head_tail(pooled, prm, keep, keep_scale, ln_eps, t)
    (out * d_out).sum().backward()
    return out.detach(), pooled.detach(), tok.grad, pooled.grad, [p.grad for p in prm], t.grad


# --------------------------------------------------------------------------------------
# CLIP
# --------------------------------------------------------------------------------------


def l2norm_scale(e: torch.Tensor, temperature: torch.Tensor) -> torch.Tensor:
    """Row L2-normalise (eps 1e-12 clamp on the norm) then multiply by exp(t/2).

    Follows modules/models.py:50-51 (``F.normalize(e, dim=-1) * torch.exp(temperature / 2)``).
    """
    norm = e.pow(2).sum(dim=-1, keepdim=True).sqrt().clamp_min(1e-12)
    return e / norm * torch.exp(temperature / 2)


def clip_logits(pep: torch.Tensor, rec: torch.Tensor) -> torch.Tensor:
    """``S = pep @ rec^T`` for ``pep [B,d]``, ``rec [B,d]`` (modules/training_utils.py:85, :260)."""
    return pep @ rec.t()


def clip_loss_from_logits(S: torch.Tensor) -> torch.Tensor:
    """Symmetric InfoNCE on a square logit matrix, written exactly as the reference does
    (no max shift): modules/training_utils.py:86-89 and :261-264."""
    E = torch.exp(S)
    d = torch.exp(torch.diagonal(S))
    L_r = -torch.mean(torch.log(d / E.sum(dim=1)))
    L_p = -torch.mean(torch.log(d / E.sum(dim=0)))
    return (L_r + L_p) * 0.5


def clip_contrastive_loss(pep: torch.Tensor, recT: torch.Tensor) -> torch.Tensor:
    """``_contrastive_loss(pep[B,d], recT[d,B])`` — modules/training_utils.py:74-89."""
    return clip_loss_from_logits(pep @ recT)


def clip_contrastive_loss_gc(xs: Sequence[torch.Tensor], ys: Sequence[torch.Tensor]) -> torch.Tensor:
    """``_contrastive_loss_gc(list, list)`` — modules/training_utils.py:247-265: the lists
    are concatenated on dim 0 (what ``@cat_input_tensor`` does), then the same arithmetic."""
    x = torch.cat(list(xs), dim=0)
    y = torch.cat(list(ys), dim=0)
    return clip_loss_from_logits(x @ y.transpose(0, 1))


def clip_get_logits(xs: Sequence[torch.Tensor], ys: Sequence[torch.Tensor]) -> torch.Tensor:
    """``get_logits(list, list)`` returns exp(x y^T) — modules/training_utils.py:157-172."""
    x = torch.cat(list(xs), dim=0)
    y = torch.cat(list(ys), dim=0)
    return torch.exp(x @ y.transpose(0, 1))


def clip_step(e_pep: torch.Tensor, e_rec: torch.Tensor, temperature: torch.Tensor):
    """Forward + backward of the whole CLIP hot path from raw head outputs:
    normalise/scale both sides (models.py:50-51), loss (training_utils.py:85-89),
    ``loss.backward()`` (training_utils.py:71).  Returns
    ``(loss, d e_pep, d e_rec, d temperature)`` by autograd."""
    e_pep = e_pep.detach().clone().requires_grad_(True)
    e_rec = e_rec.detach().clone().requires_grad_(True)
    t = temperature.detach().clone().requires_grad_(True)
    x = l2norm_scale(e_pep, t)
    y = l2norm_scale(e_rec, t)
    loss = clip_contrastive_loss(x, y.t())
    loss.backward()
    return loss.detach(), e_pep.grad, e_rec.grad, t.grad


def clip_step_analytic(e_pep: torch.Tensor, e_rec: torch.Tensor, temperature: torch.Tensor):
    """Same outputs as :func:`clip_step` from the closed-form gradient
    ``dS = (0.5/B)(softmax_rows(S) + softmax_cols(S) - 2I)``, ``dX = dS Y``, ``dY = dS^T X``,
    then the normalise/scale Jacobian.  Independent of autograd; used to referee it."""
    B = e_pep.shape[0]
    s = torch.exp(temperature / 2)
    na = e_pep.pow(2).sum(-1, keepdim=True).sqrt().clamp_min(1e-12)
    nb = e_rec.pow(2).sum(-1, keepdim=True).sqrt().clamp_min(1e-12)
    x = e_pep / na * s
    y = e_rec / nb * s
    S = x @ y.t()
    lse_r = torch.logsumexp(S, dim=1)
    lse_c = torch.logsumexp(S, dim=0)
    diag = torch.diagonal(S)
    loss = 0.5 * ((lse_r - diag).mean() + (lse_c - diag).mean())
    G = (0.5 / B) * (torch.exp(S - lse_r[:, None]) + torch.exp(S - lse_c[None, :]))
    G = G - (1.0 / B) * torch.eye(B, dtype=S.dtype)
    dx = G @ y
    dy = G.t() @ x
    # x = s * e/|e|  ->  de = (dx - n (n.dx)) * s/|e| ;   dx/dt = x/2
    n_a = e_pep / na
    n_b = e_rec / nb
    de_pep = (dx - n_a * (n_a * dx).sum(-1, keepdim=True)) * (s / na)
    de_rec = (dy - n_b * (n_b * dy).sum(-1, keepdim=True)) * (s / nb)
    dt = 0.5 * ((dx * x).sum() + (dy * y).sum())
    return loss, de_pep, de_rec, dt


# --------------------------------------------------------------------------------------
# FILIP
# --------------------------------------------------------------------------------------


def filip_normalise(h: torch.Tensor) -> torch.Tensor:
    """Per-token L2 normalise with NO epsilon — modules/models.py:238-239."""
    return h / h.pow(2).sum(dim=-1, keepdim=True).sqrt()


def _masked_mean(t: torch.Tensor, mask: torch.Tensor, dim: int, eps: float = 1e-6) -> torch.Tensor:
    """Zero the masked entries, sum, divide by the clamped count — modules/models.py:247-264."""
    t = torch.where(mask, t, torch.zeros((), dtype=t.dtype))
    denom = mask.sum(dim=dim).clamp(min=eps)
    return t.sum(dim=dim) / denom


def filip_similarity_block(hA, hB, maskA, maskB, temperature):
    """Token-wise max similarity for every (a-sequence, b-sequence) pair of one block.

    hA [bA,TA,d], hB [bB,TB,d] (already normalised), maskA [bA,TA], maskB [bB,TB] (any
    integer/bool dtype), scalar temperature (divides).  Returns sim_A, sim_B, both [bA,bB].
    Follows modules/models.py:323-356: sim/t, pad pairs -> finfo.min, max over tB per tA
    and over tA per tB, masked mean over the valid tokens of the reduced side.
    """
    mA = maskA.bool()
    mB = maskB.bool()
    sim = torch.einsum("iad,jbd->ijab", hA, hB) / temperature
    pair_ok = mA[:, None, :, None] & mB[None, :, None, :]
    sim = torch.where(pair_ok, sim, torch.full((), torch.finfo(sim.dtype).min, dtype=sim.dtype))
    rA = sim.amax(dim=3)  # [bA,bB,TA]
    rB = sim.amax(dim=2)  # [bA,bB,TB]
    sim_A = _masked_mean(rA, mA[:, None, :].expand_as(rA), dim=-1)
    sim_B = _masked_mean(rB, mB[None, :, :].expand_as(rB), dim=-1)
    return sim_A, sim_B


def filip_similarity(hA, hB, maskA, maskB, temperature, block: int = 16):
    """Full [BA,BB] score matrices computed block by block; exact because the scores of a
    pair depend on that pair alone (so the [BA,BB,TA,TB] tensor of models.py:332 is never
    held whole)."""
    BA, BB = hA.shape[0], hB.shape[0]
    sim_A = torch.empty(BA, BB, dtype=hA.dtype)
    sim_B = torch.empty(BA, BB, dtype=hA.dtype)
    for i0 in range(0, BA, block):
        for j0 in range(0, BB, block):
            a, b = filip_similarity_block(
                hA[i0:i0 + block], hB[j0:j0 + block], maskA[i0:i0 + block], maskB[j0:j0 + block], temperature
            )
            sim_A[i0:i0 + block, j0:j0 + block] = a
            sim_B[i0:i0 + block, j0:j0 + block] = b
    return sim_A, sim_B


def row_ce(logits: torch.Tensor, use_dcl: bool = False) -> torch.Tensor:
    """Row-wise InfoNCE with the diagonal as target, as written in
    modules/training_utils_filip.py:4-23 (``use_dcl`` removes the diagonal from the
    denominator)."""
    E = logits.exp()
    num = torch.diagonal(E)
    if use_dcl:
        E = E - torch.diag(torch.diagonal(E))
    den = E.sum(dim=1)
    return (-num.log() + den.log()).mean()


def filip_loss(sim_A: torch.Tensor, sim_B: torch.Tensor) -> torch.Tensor:
    """``_compute_loss`` — modules/training_utils_filip.py:26-39."""
    return (row_ce(sim_A) + row_ce(sim_B)) / 2


def filip_step(hA_raw, hB_raw, maskA, maskB, temperature, block: int = 16):
    """Forward + backward of the whole FILIP hot path from raw token embeddings:
    normalise (models.py:238-239), scores (models.py:292-356), loss
    (training_utils_filip.py:26-39), ``loss.backward()`` (training_utils_filip.py:63).

    Two passes so memory stays bounded at any batch: pass 1 builds the score matrices block
    by block without a graph; the loss and d loss/d scores come from those; pass 2 re-runs
    each block with a graph and pushes its slice of d loss/d scores through it.
    Returns ``(loss, sim_A, sim_B, d hA_raw, d hB_raw, d temperature)``.
    """
    BA, BB = hA_raw.shape[0], hB_raw.shape[0]
    t = temperature.detach().clone()
    with torch.no_grad():
        sA, sB = filip_similarity(filip_normalise(hA_raw), filip_normalise(hB_raw), maskA, maskB, t, block)
    sA_l = sA.clone().requires_grad_(True)
    sB_l = sB.clone().requires_grad_(True)
    loss = filip_loss(sA_l, sB_l)
    loss.backward()
    gA, gB = sA_l.grad, sB_l.grad
    dA = torch.zeros_like(hA_raw)
    dB = torch.zeros_like(hB_raw)
    dt = torch.zeros((), dtype=hA_raw.dtype)
    for i0 in range(0, BA, block):
        for j0 in range(0, BB, block):
            a = hA_raw[i0:i0 + block].detach().clone().requires_grad_(True)
            b = hB_raw[j0:j0 + block].detach().clone().requires_grad_(True)
            tt = t.clone().requires_grad_(True)
            s1, s2 = filip_similarity_block(
                filip_normalise(a), filip_normalise(b), maskA[i0:i0 + block], maskB[j0:j0 + block], tt
            )
            obj = (s1 * gA[i0:i0 + block, j0:j0 + block]).sum() + (s2 * gB[i0:i0 + block, j0:j0 + block]).sum()
            obj.backward()
            dA[i0:i0 + block] += a.grad
            dB[j0:j0 + block] += b.grad
            dt += tt.grad
    return loss.detach(), sA, sB, dA, dB, dt


# --------------------------------------------------------------------------------------
# helpers shared by tests and the bench
# --------------------------------------------------------------------------------------


def bf16_round(x: torch.Tensor) -> torch.Tensor:
    """Round to the nearest bf16 value and return it in the input dtype (so a CPU float32
    computation and a bf16 matrix-core computation see identical operands)."""
    return x.to(torch.bfloat16).to(x.dtype)


def right_padded_masks(lengths: torch.Tensor, T: int) -> torch.Tensor:
    """uint8 [B,T] mask with ones in the first ``lengths[b]`` positions of row b."""
    return (torch.arange(T)[None, :] < lengths[:, None]).to(torch.uint8)
