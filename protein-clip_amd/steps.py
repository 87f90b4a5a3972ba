"""Explicit forward+backward steps of the two hot paths, single-GPU or sharded over a
process group (one process per GPU, RCCL), without autograd in the way.  These are what
``bench.py`` times and what a trainer that manages its own gradients would call.

CLIP (replaces GradCache's accumulation, modules/training_utils.py:117-153): every rank
normalises its own rows, ONE all-gather moves both towers' embeddings, the rank scores its
row block and its column block of the global logits, ONE all-gather moves the two
log-sum-exp vectors, the backward produces gradients for the rank's own rows only (no
gradient all-reduce of embeddings), ONE all-reduce sums {loss, dt}.

FILIP (modules/models.py:221-245 + modules/training_utils_filip.py:26-39): both losses are
row-wise, so a rank that owns A rows needs every B sequence: all-gather of the bf16 token
images (+ mask words), local scores / losses / dA, reduce-scatter of dB, one all-reduce of
{loss_A, loss_B, dt}.
"""
from __future__ import annotations

import torch
import torch.distributed as dist

from . import kernels as _hip_kernels


# Tests set this to run every collective in a one-rank process group as well (the GPU box has
# one GPU; RCCL accepts a group of one), so that dtypes and shapes meet the real backend.
FORCE_COLLECTIVES = False


def _collective(world: int) -> bool:
    return world > 1 or (FORCE_COLLECTIVES and dist.is_available() and dist.is_initialized())


def _world(group):
    if group is None and not (dist.is_available() and dist.is_initialized()):
        return 1, 0
    return dist.get_world_size(group), dist.get_rank(group)


def _gather(t, group, world):
    if not _collective(world):
        return t
    # concatenated-along-dim-0 output is the layout every backend (RCCL, gloo) accepts
    out = torch.empty((world * t.shape[0],) + tuple(t.shape[1:]), dtype=t.dtype, device=t.device)
    src = t.contiguous()
    if t.dtype in (torch.float32, torch.float64, torch.float16, torch.bfloat16):
        dist.all_gather_into_tensor(out, src, group=group)
    else:
        # a gather only moves bytes, and RCCL has no 16-bit integer type (the bf16 images travel
        # as int16 bit patterns): send every integer tensor as bytes
        dist.all_gather_into_tensor(out.view(torch.uint8), src.view(torch.uint8), group=group)
    return out.view((world,) + tuple(t.shape))


class ClipShardedStep:
    def __init__(self, group=None, kernels=None):
        self.group, self.K = group, kernels or _hip_kernels
        self.world, self.rank = _world(group)

    def run(self, e_pep, e_rec, t):
        """e_pep/e_rec [Bl,d] f32: this rank's raw head outputs; t: 1-element f32 tensor.
        Returns (loss, d e_pep, d e_rec, dt) — loss and dt are global (identical on all ranks)."""
        K, W = self.K, self.world
        Bl, d = e_pep.shape
        x = K.l2norm_scale_fwd(e_pep, t)
        y = K.l2norm_scale_fwd(e_rec, t)
        if _collective(W):
            both = _gather(torch.stack([x, y]), self.group, W)       # [W,2,Bl,d]
            x_all = both[:, 0].reshape(W * Bl, d).contiguous()
            y_all = both[:, 1].reshape(W * Bl, d).contiguous()
        else:
            x_all, y_all = x, y
        off = self.rank * Bl
        lse_r, lse_c, loss, _ = K.clip_fwd(x, y, x_all, y_all, off)
        if _collective(W):
            lse = _gather(torch.stack([lse_r, lse_c]), self.group, W)  # [W,2,Bl]
            lse_r_all = lse[:, 0].reshape(-1).contiguous()
            lse_c_all = lse[:, 1].reshape(-1).contiguous()
        else:
            lse_r_all, lse_c_all = lse_r, lse_c
        dX, dY = K.clip_bwd(x, y, x_all, y_all, lse_r_all, lse_c_all, off)
        de_pep, dt = K.l2norm_scale_bwd(e_pep, t, dX)
        de_rec, dt = K.l2norm_scale_bwd(e_rec, t, dY, dt=dt)
        if _collective(W):
            pack = torch.cat([loss.reshape(1), dt.reshape(1)])
            dist.all_reduce(pack, group=self.group)
            loss, dt = pack[0:1], pack[1:2]
        return loss, de_pep, de_rec, dt


class FilipShardedStep:
    def __init__(self, group=None, kernels=None):
        self.group, self.K = group, kernels or _hip_kernels
        self.world, self.rank = _world(group)
        self.last = {}

    def run(self, hA, hB, maskA, maskB, t, sim_events=None):
        """hA [Bl,TA,d], hB [Bl,TB,d] f32 raw token embeddings of this rank's pairs; masks
        [Bl,T]; t 1-element f32 tensor.  Returns (loss, d hA, d hB, dt)."""
        K, W = self.K, self.world
        Bl, TA, TB = hA.shape[0], hA.shape[1], hB.shape[1]
        Bg = W * Bl
        hnA, mwA = K.filip_prep(hA, maskA, True, tsign=t)
        hnB, mwB = K.filip_prep(hB, maskB, True)
        if _collective(W):
            hnB_all = _gather(hnB, self.group, W).reshape((Bg,) + tuple(hnB.shape[1:]))
            mwB_all = _gather(mwB, self.group, W).reshape((Bg,) + tuple(mwB.shape[1:]))
        else:
            hnB_all, mwB_all = hnB, mwB
        if sim_events is not None:
            sim_events[0].record()
        simA, simB, argA, argB = K.filip_sim_fwd(hnA, mwA, TA, hnB_all, mwB_all, TB, t, want_arg=True, unit_norm=True)
        if sim_events is not None:
            sim_events[1].record()
        off = self.rank * Bl
        loss3, gA, gB = K.filip_loss(simA, simB, off, Bg)   # [(lossA + lossB) / 2, lossA, lossB], d loss / d sim
        if _collective(W):
            # B side first: its gradient is the one that travels (every rank holds a share for every
            # B sequence), and the reduce-scatter then runs beside the A side's kernels
            _, dnB_all, _ = K.filip_sim_bwd(hnA, mwA, TA, hnB_all, mwB_all, TB, t, simA, simB, argA, argB, gA, gB, sides="B")
            dnB = torch.empty((Bl,) + tuple(dnB_all.shape[1:]), dtype=dnB_all.dtype, device=dnB_all.device)
            work = dist.reduce_scatter_tensor(dnB, dnB_all, group=self.group, async_op=True)
            dnA, _, dt = K.filip_sim_bwd(hnA, mwA, TA, hnB_all, mwB_all, TB, t, simA, simB, argA, argB, gA, gB, sides="A", resume=True)
            dhA = K.filip_prep_bwd(hA, dnA, True, tsign=t)
            work.wait()
            dhB = K.filip_prep_bwd(hB, dnB, True)
        else:   # one GPU: no exchange between the two backward stages, so they run fused
            dhA, dhB, dt = K.filip_bwd_fused(hnA, mwA, hnB_all, mwB_all, t, simA, simB, argA, argB, gA, gB, hA, hB, True)
        self.last = {"simA": simA, "simB": simB}
        if _collective(W):
            pack = torch.cat([loss3[0:1], dt.reshape(1)])
            dist.all_reduce(pack, group=self.group)
            return pack[0:1], dhA, dhB, pack[1:2]
        return loss3[0:1], dhA, dhB, dt.reshape(1)
