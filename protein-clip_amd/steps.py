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


def _gather_into(out, src, group, async_op=False):
    """All-gather of ``src`` into ``out`` ([world * n, ...], rank-major).  A gather only moves bytes, and RCCL has no
    16-bit integer type (the bf16 images travel as int16 bit patterns): integer tensors are sent as bytes."""
    if src.dtype in (torch.float32, torch.float64, torch.float16, torch.bfloat16):
        return dist.all_gather_into_tensor(out, src, group=group, async_op=async_op)
    return dist.all_gather_into_tensor(out.view(torch.uint8), src.view(torch.uint8), group=group, async_op=async_op)


class ClipShardedStep:
    """Every device buffer of the step (outputs, gathered batches, scratch) is created on the first ``run`` and used
    again by every later one with the same shapes (``kernels.BufferArena``): a step allocates nothing.  What ``run``
    returns is overwritten by the next ``run``."""

    def __init__(self, group=None, kernels=None):
        self.group, self.K = group, kernels or _hip_kernels
        self.world, self.rank = _world(group)
        self.arena = _hip_kernels.BufferArena()

    def run(self, e_pep, e_rec, t):
        """e_pep/e_rec [Bl,d] f32: this rank's raw head outputs; t: 1-element f32 tensor.
        Returns (loss, d e_pep, d e_rec, dt) — loss and dt are global (identical on all ranks)."""
        K, W = self.K, self.world
        Bl, d = e_pep.shape
        dev, f32 = e_pep.device, torch.float32
        with self.arena as A:
            if _collective(W):
                # both towers travel in one collective; the kernels want each tower's gathered rows in one piece
                src = A.take((2, Bl, d), f32, dev)
                x = K.l2norm_scale_fwd(e_pep, t, out=src[0])
                y = K.l2norm_scale_fwd(e_rec, t, out=src[1])
                both = A.take((W, 2, Bl, d), f32, dev)
                _gather_into(both.view(W * 2 * Bl, d), src.view(2 * Bl, d), self.group)
                x_all, y_all = A.take((W * Bl, d), f32, dev), A.take((W * Bl, d), f32, dev)
                x_all.view(W, Bl, d).copy_(both[:, 0])
                y_all.view(W, Bl, d).copy_(both[:, 1])
            else:
                x = K.l2norm_scale_fwd(e_pep, t)
                y = K.l2norm_scale_fwd(e_rec, t)
                x_all, y_all = x, y
            off = self.rank * Bl
            lse_r, lse_c, loss, _ = K.clip_fwd(x, y, x_all, y_all, off)
            if _collective(W):
                lsrc = A.take((2, Bl), f32, dev)
                lsrc[0].copy_(lse_r)
                lsrc[1].copy_(lse_c)
                lse = A.take((W, 2, Bl), f32, dev)
                _gather_into(lse.view(W * 2, Bl), lsrc, self.group)
                lse_r_all, lse_c_all = A.take((W * Bl,), f32, dev), A.take((W * Bl,), f32, dev)
                lse_r_all.view(W, Bl).copy_(lse[:, 0])
                lse_c_all.view(W, Bl).copy_(lse[:, 1])
            else:
                lse_r_all, lse_c_all = lse_r, lse_c
            dX, dY = K.clip_bwd(x, y, x_all, y_all, lse_r_all, lse_c_all, off)
            de_pep, dt = K.l2norm_scale_bwd(e_pep, t, dX)
            de_rec, dt = K.l2norm_scale_bwd(e_rec, t, dY, dt=dt)
            if _collective(W):
                pack = A.take((2,), f32, dev)
                pack[0:1].copy_(loss.reshape(1))
                pack[1:2].copy_(dt.reshape(1))
                dist.all_reduce(pack, group=self.group)
                loss, dt = pack[0:1], pack[1:2]
        return loss, de_pep, de_rec, dt


class FilipShardedStep:
    """As :class:`ClipShardedStep`: nothing is allocated after the first ``run`` of a shape.  The all-gather of the
    other ranks' B sequences runs beside the scoring of the rank's own block (the scores of a column block are the
    same bits whichever call computes them), the reduce-scatter of the B gradients beside the A side's backward."""

    def __init__(self, group=None, kernels=None):
        self.group, self.K = group, kernels or _hip_kernels
        self.world, self.rank = _world(group)
        self.arena = _hip_kernels.BufferArena()
        self.last = {}

    def run(self, hA, hB, maskA, maskB, t, sim_events=None):
        """hA [Bl,TA,d], hB [Bl,TB,d] f32 raw token embeddings of this rank's pairs; masks
        [Bl,T]; t 1-element f32 tensor.  Returns (loss, d hA, d hB, dt)."""
        K, W, rank = self.K, self.world, self.rank
        Bl, TA, TB = hA.shape[0], hA.shape[1], hB.shape[1]
        Bg = W * Bl
        dev = hA.device
        with self.arena as A:
            hnA, mwA = K.filip_prep(hA, maskA, True, tsign=t)
            hnB, mwB = K.filip_prep(hB, maskB, True)
            if sim_events is not None:
                sim_events[0].record()
            if _collective(W):
                hnB_all = A.take((Bg,) + tuple(hnB.shape[1:]), hnB.dtype, dev)
                mwB_all = A.take((Bg,) + tuple(mwB.shape[1:]), mwB.dtype, dev)
                works = [_gather_into(hnB_all, hnB.contiguous(), self.group, async_op=True),
                         _gather_into(mwB_all, mwB.contiguous(), self.group, async_op=True)]
                TAp, TBp = _round32(TA), _round32(TB)
                out = (A.take((Bl, Bg), torch.float32, dev), A.take((Bl, Bg), torch.float32, dev),
                       A.take((Bl, Bg, TAp), torch.int16, dev), A.take((Bl, Bg, TBp), torch.int16, dev))
                # the rank's own B sequences while the others are on their way, then the blocks in front of and behind them
                K.filip_sim_fwd(hnA, mwA, TA, hnB, mwB, TB, t, want_arg=True, unit_norm=True, out=out, col_offset=rank * Bl)
                for wk in works:
                    wk.wait()
                if rank > 0:
                    K.filip_sim_fwd(hnA, mwA, TA, hnB_all[:rank * Bl], mwB_all[:rank * Bl], TB, t, want_arg=True, unit_norm=True, out=out,
                                    col_offset=0)
                if rank < W - 1:
                    K.filip_sim_fwd(hnA, mwA, TA, hnB_all[(rank + 1) * Bl:], mwB_all[(rank + 1) * Bl:], TB, t, want_arg=True, unit_norm=True,
                                    out=out, col_offset=(rank + 1) * Bl)
                simA, simB, argA, argB = out
            else:
                hnB_all, mwB_all = hnB, mwB
                simA, simB, argA, argB = K.filip_sim_fwd(hnA, mwA, TA, hnB_all, mwB_all, TB, t, want_arg=True, unit_norm=True)
            if sim_events is not None:
                sim_events[1].record()
            off = rank * Bl
            loss3, gA, gB = K.filip_loss(simA, simB, off, Bg)   # [(lossA + lossB) / 2, lossA, lossB], d loss / d sim
            if _collective(W):
                # B side first: its gradient is the one that travels (every rank holds a share for every
                # B sequence), and the reduce-scatter then runs beside the A side's kernels.  The two one-sided calls
                # share a workspace of this step's own: the second finds the first one's marks of mutual choices in it.
                ws = K.filip_bwd_workspace(Bl, TA, Bg, TB, dev)
                _, dnB_all, _ = K.filip_sim_bwd(hnA, mwA, TA, hnB_all, mwB_all, TB, t, simA, simB, argA, argB, gA, gB, sides="B", ws=ws)
                dnB = A.take((Bl,) + tuple(dnB_all.shape[1:]), dnB_all.dtype, dev)
                work = dist.reduce_scatter_tensor(dnB, dnB_all, group=self.group, async_op=True)
                dnA, _, dt = K.filip_sim_bwd(hnA, mwA, TA, hnB_all, mwB_all, TB, t, simA, simB, argA, argB, gA, gB, sides="A", resume=True,
                                             ws=ws)
                dhA = K.filip_prep_bwd(hA, dnA, True, tsign=t)
                work.wait()
                dhB = K.filip_prep_bwd(hB, dnB, True)
            else:   # one GPU: no exchange between the two backward stages, so they run fused
                dhA, dhB, dt = K.filip_bwd_fused(hnA, mwA, hnB_all, mwB_all, t, simA, simB, argA, argB, gA, gB, hA, hB, True)
            self.last = {"simA": simA, "simB": simB}
            if _collective(W):
                pack = A.take((2,), torch.float32, dev)
                pack[0:1].copy_(loss3[0:1])
                pack[1:2].copy_(dt.reshape(1))
                dist.all_reduce(pack, group=self.group)
                return pack[0:1], dhA, dhB, pack[1:2]
        return loss3[0:1], dhA, dhB, dt.reshape(1)


def _round32(T: int) -> int:
    return (T + 31) // 32 * 32
