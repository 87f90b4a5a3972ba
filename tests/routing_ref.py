"""TEST-ONLY reference of the FILIP score backward GIVEN the arg-max tables: the max routes every
gradient to one (a, b) token pair (autograd of models.py:332-351 of the reference), so with the
routing fixed the result is a plain weighted sum of operand rows.  Written with torch index
operations in float64; runs wherever its inputs live.  It lets a test separate the two things a
backward kernel can get wrong: which token a gradient goes to (checked by value: the chosen token
attains the maximum) and the arithmetic once that is settled (checked here, tightly)."""
import torch


def decode_image(hn: torch.Tensor, T: int) -> torch.Tensor:
    """[B,Tp,128] int16 bf16 bit patterns (pclip_filip_prep output) -> [B,T,128] float64."""
    return hn.view(torch.bfloat16)[:, :T].to(torch.float64)


def routed_grads(nA, nB, mA, mB, argA, argB, gA, gB, abs_t: float):
    """nA [Br,TA,d], nB [Bc,TB,d]: the operands the matrix cores saw (float64); mA/mB [.,T] bool;
    argA [Br,Bc,>=TA], argB [Br,Bc,>=TB]: the partner token chosen by every token (entries of
    padding tokens are ignored); gA/gB [Br,Bc] = d loss / d sim_A, d sim_B.  Scores are
    (a . b) / abs_t.  Returns (dnA, dnB) = d loss / d operands."""
    Br, TA, d = nA.shape
    Bc, TB, _ = nB.shape
    mA, mB = mA.bool(), mB.bool()
    cntA, cntB = mA.sum(1).double(), mB.sum(1).double()
    liveA, liveB = cntA > 0, cntB > 0
    wA = torch.where(liveB[None, :], gA.double() / cntA.clamp(min=1e-6)[:, None] / abs_t, torch.zeros((), dtype=torch.float64, device=nA.device))
    wB = gB.double() / cntB.clamp(min=1e-6)[None, :] / abs_t
    dA, dB = torch.zeros_like(nA), torch.zeros_like(nB)
    jj = torch.arange(Bc, device=nA.device)[:, None]
    for i in range(Br):
        if not bool(liveA[i]):
            continue
        # tokens of A sequence i and the B token each of them chose, for every B sequence j
        b = argA[i, :, :TA].long().clamp(0, TB - 1)                       # [Bc,TA]
        w = wA[i][:, None] * mA[i][None, :]                                # [Bc,TA]
        dA[i] += (w[..., None] * nB[jj, b]).sum(0)
        dB.index_put_((jj.expand(Bc, TA), b), w[..., None] * nA[i][None], accumulate=True)
        # tokens of every B sequence j and the token of A sequence i each of them chose
        a = argB[i, :, :TB].long().clamp(0, TA - 1)                       # [Bc,TB]
        w2 = wB[i][:, None] * mB                                           # [Bc,TB]
        dB += w2[..., None] * nA[i][a]
        dA[i].index_add_(0, a.reshape(-1), (w2[..., None] * nB).reshape(-1, d))
    return dA, dB


def chosen_values(nA, nB, argA, argB, i_rows):
    """For the A sequences in ``i_rows``: the similarity each token reaches with its chosen partner
    and the true maximum, both directions: (pickA, maxA [n,Bc,TA], pickB, maxB [n,Bc,TB]); masks
    are the caller's business."""
    TA, TB = nA.shape[1], nB.shape[1]
    sim = torch.einsum("iad,jbd->ijab", nA[i_rows], nB)
    pa = torch.gather(sim, 3, argA[i_rows][:, :, :TA, None].long().clamp(0, TB - 1)).squeeze(3)
    pb = torch.gather(sim, 2, argB[i_rows][:, :, None, :TB].long().clamp(0, TA - 1)).squeeze(2)
    return pa, sim, pb
