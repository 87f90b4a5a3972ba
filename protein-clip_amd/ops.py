"""Autograd operators over the HIP kernels, single-GPU and sharded (one process per GPU,
``torch.distributed`` over RCCL).

The sharded forms replace GradCache's sub-batch accumulation
(modules/training_utils.py:92-155 in the reference): every rank encodes its own rows, the
embeddings are all-gathered, each rank scores its row block of the global similarity matrix
and produces gradients for its own rows only, so the result equals the single-device
global-batch loss.

All compute goes through a *kernel provider* (default: :mod:`protein_clip_amd.kernels`, the
ctypes binding of libpclip.so).  There is no CPU implementation in this package; CPU
tensors are rejected by the provider.  The provider argument exists so the communication
logic can be exercised on CPU by tests that bring their own stand-in.
"""
from __future__ import annotations

from typing import Optional

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


def _all_gather_rows(t: torch.Tensor, group, world: int) -> torch.Tensor:
    """[n, ...] on every rank -> [world*n, ...], rank-major (rank r's rows at r*n)."""
    if not _collective(world):
        return t
    out = torch.empty((world * t.shape[0],) + tuple(t.shape[1:]), dtype=t.dtype, device=t.device)
    src = t.contiguous()
    if t.dtype in (torch.float32, torch.float64, torch.float16, torch.bfloat16):
        dist.all_gather_into_tensor(out, src, group=group)
    else:
        # a gather only moves bytes, and RCCL has no 16-bit integer type (the bf16 images travel
        # as int16 bit patterns): send every integer tensor as bytes
        dist.all_gather_into_tensor(out.view(torch.uint8), src.view(torch.uint8), group=group)
    return out


# --------------------------------------------------------------------------------------
# L2 normalise + temperature scale   (modules/models.py:50-51)
# --------------------------------------------------------------------------------------
class _L2NormScale(torch.autograd.Function):
    @staticmethod
    def forward(ctx, e, t, K):
        e = e.contiguous()
        ctx.K = K
        ctx.save_for_backward(e, t)
        return K.l2norm_scale_fwd(e, t)

    @staticmethod
    def backward(ctx, dx):
        e, t = ctx.saved_tensors
        de, dt = ctx.K.l2norm_scale_bwd(e, t, dx.contiguous())
        return de, dt.reshape(t.shape).to(t.dtype), None


def _f32(t: torch.Tensor) -> torch.Tensor:
    """Half / bfloat16 activations (a caller running under autocast, as the reference's ``train_gc``
    does) are upcast here, through autograd; the kernels compute in f32 (bf16 MFMA operands for
    FILIP) whatever the caller's precision context is."""
    return t if t.dtype == torch.float32 or not t.is_floating_point() else t.float()


class _MaskedMean(torch.autograd.Function):
    @staticmethod
    def forward(ctx, h, mask, K):
        mean, inv, m8 = K.masked_mean_fwd(h.contiguous(), mask)
        ctx.save_for_backward(m8, inv)
        ctx.K, ctx.T = K, h.shape[1]
        return mean

    @staticmethod
    def backward(ctx, g):
        m8, inv = ctx.saved_tensors
        return ctx.K.masked_mean_bwd(g.contiguous(), m8, inv, ctx.T), None, None


def masked_mean(h: torch.Tensor, mask: torch.Tensor, kernels=None) -> torch.Tensor:
    """``sum_t h[b,t,:] * mask[b,t] / max(sum_t mask[b,t], 1)`` — Encoder._masked_mean of the
    reference (modules/models.py:76-90) as one pass over ``h`` (and one in the backward)."""
    if h.dim() != 3 or tuple(mask.shape) != tuple(h.shape[:2]):
        raise ValueError("masked_mean: expected h [B,T,d] and mask [B,T]")
    return _MaskedMean.apply(_f32(h), mask, kernels or _hip_kernels)


# limits of the fused embedding-head kernel (PCLIP_HEAD_MAX_BLOCKS, and its width, in include/pclip.h)
HEAD_MAX_BLOCKS = _hip_kernels.HEAD_MAX_BLOCKS
HEAD_MAX_D = _hip_kernels.HEAD_MAX_D


class _HeadTail(torch.autograd.Function):
    @staticmethod
    def forward(ctx, pooled, t, keep, keep_scale, ln_eps, K, *params):
        pooled = pooled.contiguous()
        params = [prm.detach().contiguous() for prm in params]
        out, save = K.head_tail_fwd(pooled, params, keep, keep_scale, ln_eps, t)
        ctx.K, ctx.keep, ctx.keep_scale, ctx.ln_eps, ctx.n = K, keep, keep_scale, ln_eps, len(params)
        ctx.save_for_backward(pooled, t, save, *params)
        return out

    @staticmethod
    def backward(ctx, dout):
        pooled, t, save = ctx.saved_tensors[:3]
        params = list(ctx.saved_tensors[3:])
        dpooled, dparams, dt = ctx.K.head_tail_bwd(pooled, params, ctx.keep, ctx.keep_scale, ctx.ln_eps, t, save, dout.contiguous())
        return (dpooled, dt.reshape(t.shape).to(t.dtype), None, None, None, None) + tuple(dparams)


def embedding_head(pooled: torch.Tensor, params, temperature: torch.Tensor, keep: Optional[torch.Tensor] = None, keep_scale: float = 1.0,
                   ln_eps: float = 1e-5, kernels=None) -> torch.Tensor:
    """The embedding head of a CLIP tower after the pooling (modules/models.py:49-51 with
    ``embedding_ffn`` as built at :54-73) as one launch forward and two backward:
    ``params`` = ``[W, b, gamma, beta]`` of each Linear -> ReLU -> LayerNorm -> Dropout block, then
    ``[W, b]`` of the last Linear; ``keep`` = the dropout masks ``[blocks, B, d]`` (uint8, None in
    eval mode) with ``keep_scale = 1 / (1 - p)``.  Returns ``normalize(z) * exp(temperature / 2)``."""
    return _HeadTail.apply(_f32(pooled), _f32(temperature), keep, float(keep_scale), float(ln_eps), kernels or _hip_kernels, *params)


def l2norm_scale(e: torch.Tensor, temperature: torch.Tensor, kernels=None) -> torch.Tensor:
    """``F.normalize(e, dim=-1) * torch.exp(temperature / 2)`` as one kernel each way."""
    return _L2NormScale.apply(_f32(e), _f32(temperature), kernels or _hip_kernels)


# --------------------------------------------------------------------------------------
# CLIP symmetric InfoNCE   (modules/training_utils.py:85-89, :260-264)
# --------------------------------------------------------------------------------------
class _ClipLoss(torch.autograd.Function):
    """loss(x_own, y_own) for the global batch formed by all ranks' rows."""

    @staticmethod
    def forward(ctx, x, y, group, K):
        world, rank = _world(group)
        x = x.contiguous()
        y = y.contiguous()
        Bl = x.shape[0]
        ctx.fused = False
        if not _collective(world) and (ctx.needs_input_grad[0] or ctx.needs_input_grad[1]) and hasattr(K, "clip_loss_grad"):
            # one process, gradients wanted: loss and both gradients come out of one call (one launch up to
            # B = 512); the backward only scales them by the upstream gradient
            loss, dx, dy = K.clip_loss_grad(x, y)
            ctx.fused = True
            ctx.save_for_backward(dx, dy)
            return loss.reshape(())
        x_all = _all_gather_rows(x, group, world)
        y_all = _all_gather_rows(y, group, world)
        off = rank * Bl
        lse_r, lse_c, loss, _ = K.clip_fwd(x, y, x_all, y_all, off)
        if _collective(world):
            # both log-sum-exp vectors travel in one collective: [2,Bl] per rank -> [world,2,Bl]
            both = torch.stack([lse_r, lse_c])
            gathered = _all_gather_rows(both, group, world).view((world,) + tuple(both.shape))
            lse_r_all = gathered[:, 0].reshape(-1).contiguous()
            lse_c_all = gathered[:, 1].reshape(-1).contiguous()
            dist.all_reduce(loss, group=group)
        else:
            lse_r_all, lse_c_all = lse_r, lse_c
        ctx.K, ctx.off = K, off
        ctx.save_for_backward(x, y, x_all, y_all, lse_r_all, lse_c_all)
        return loss.reshape(())

    @staticmethod
    def backward(ctx, g):
        if ctx.fused:
            dx, dy = ctx.saved_tensors
            return dx * g, dy * g, None, None
        x, y, x_all, y_all, lse_r_all, lse_c_all = ctx.saved_tensors
        dX, dY = ctx.K.clip_bwd(x, y, x_all, y_all, lse_r_all, lse_c_all, ctx.off, g=g.reshape(1))
        return dX, dY, None, None


def clip_loss(x: torch.Tensor, y: torch.Tensor, group=None, kernels=None) -> torch.Tensor:
    """Symmetric InfoNCE of ``S = X_all Y_all^T`` where X_all/Y_all are the rows of every rank
    (just ``x``/``y`` without a process group).  Every rank returns the same global loss; its
    backward yields gradients for the rank's own rows."""
    if x.dim() != 2 or x.shape != y.shape:
        raise ValueError(f"clip_loss: expected two [B,d] tensors of one shape, got {tuple(x.shape)} and {tuple(y.shape)}")
    return _ClipLoss.apply(_f32(x), _f32(y), group, kernels or _hip_kernels)


def clip_exp_logits(x: torch.Tensor, y: torch.Tensor, kernels=None) -> torch.Tensor:
    """``exp(x @ y.T)`` — what ``get_logits`` returns (modules/training_utils.py:170-172)."""
    K = kernels or _hip_kernels
    x = _f32(x.detach()).contiguous()
    y = _f32(y.detach()).contiguous()
    return K.clip_fwd(x, y, x, y, 0, logits_mode=2)[3]


# --------------------------------------------------------------------------------------
# row-wise InfoNCE   (modules/training_utils_filip.py:4-23)
# --------------------------------------------------------------------------------------
class _RowCE(torch.autograd.Function):
    @staticmethod
    def forward(ctx, logits, use_dcl, row_offset, Bg, group, K):
        logits = logits.contiguous()
        lse, loss = K.rowce_fwd(logits, row_offset, Bg, use_dcl)
        if group is not False:   # False: a whole matrix on this rank, nothing to sum over ranks
            world, _ = _world(group)
            if _collective(world):
                dist.all_reduce(loss, group=group)
        ctx.K, ctx.cfg = K, (use_dcl, row_offset, Bg)
        ctx.save_for_backward(logits, lse)
        return loss.reshape(())

    @staticmethod
    def backward(ctx, g):
        logits, lse = ctx.saved_tensors
        use_dcl, row_offset, Bg = ctx.cfg
        return ctx.K.rowce_bwd(logits, lse, row_offset, Bg, use_dcl, g=g.reshape(1)), None, None, None, None, None


# Off by default: scores that :func:`filip_similarity` computed under a process group carry a tag saying which rows of
# the global matrix they are, and only tagged tensors are treated as row blocks.  Set to True to also take any
# ``[Bl, W*Bl]`` matrix under an initialised group of W ranks for this rank's row block (a guess from the shape alone).
INFER_ROW_SHARD_FROM_SHAPE = False
_ROW_SHARD_TAG = "_pclip_row_shard"


def _row_shard(logits: torch.Tensor, group):
    """(row_offset, global_batch, sharded) of a matrix of scores.  What :func:`filip_similarity` returns under a process
    group of W > 1 ranks — this rank's ``Bl`` A rows against the gathered ``W*Bl`` B sequences — is tagged with its row
    offset ``rank*Bl`` and the global batch: the target of local row i is column ``rank*Bl + i`` and the mean runs over
    the rows of all ranks.  An untagged matrix is a complete one of its own: (0, rows, False)."""
    tag = getattr(logits, _ROW_SHARD_TAG, None)
    if tag is not None and tuple(logits.shape) == tag[2]:
        return tag[0], tag[1], True
    Bl, Bc = logits.shape
    if INFER_ROW_SHARD_FROM_SHAPE:
        world, rank = _world(group)
        if world > 1 and Bc == world * Bl:
            return rank * Bl, Bc, True
    return 0, Bl, False


def row_ce(logits: torch.Tensor, use_dcl: bool = False, row_offset: Optional[int] = None, global_batch: Optional[int] = None, group=None,
           kernels=None) -> torch.Tensor:
    """``mean_i(-log E_ii + log sum_j E_ij)`` over the global batch; ``logits`` holds this rank's
    rows, whose targets sit at column ``i + row_offset``.

    Without explicit ``row_offset`` / ``global_batch``: scores that :func:`filip_similarity` computed under a
    process group say themselves which row block of the global matrix they are (the loss is then summed
    over the ranks, every rank returns the global value and back-propagates its own rows); anything else
    is a whole matrix with the diagonal as target."""
    if logits.dim() != 2:
        raise ValueError("row_ce: expected a 2-d logits tensor")
    if row_offset is None and global_batch is None:
        row_offset, Bg, sharded = _row_shard(logits, group)
    else:
        row_offset = 0 if row_offset is None else row_offset
        Bg = logits.shape[0] if global_batch is None else global_batch
        sharded = Bg != logits.shape[0]
    return _RowCE.apply(_f32(logits), bool(use_dcl), row_offset, Bg, group if (sharded or FORCE_COLLECTIVES) else False, kernels or _hip_kernels)


# --------------------------------------------------------------------------------------
# FILIP scores   (modules/models.py:238-239, :292-356)
# --------------------------------------------------------------------------------------
class _FilipSim(torch.autograd.Function):
    @staticmethod
    def forward(ctx, hA, hB, maskA, maskB, t, normalise, group, K):
        world, rank = _world(group)
        hA = hA.contiguous()
        hB = hB.contiguous()
        # the A image is negated on the device when t < 0 (max(x/t) = max(-x)/|t|): no host read of t
        hnA, mwA = K.filip_prep(hA, maskA, normalise, tsign=t)
        hnB, mwB = K.filip_prep(hB, maskB, normalise)
        hnB_all = _all_gather_rows(hnB, group, world)
        mwB_all = _all_gather_rows(mwB, group, world)
        TA, TB = hA.shape[1], hB.shape[1]
        simA, simB, argA, argB = K.filip_sim_fwd(hnA, mwA, TA, hnB_all, mwB_all, TB, t, want_arg=True, unit_norm=bool(normalise))
        ctx.K, ctx.cfg = K, (normalise, TA, TB, group, world, rank)
        ctx.save_for_backward(hA, hB, hnA, mwA, hnB_all, mwB_all, t, simA, simB, argA, argB)
        ctx.mark_non_differentiable(argA, argB)
        return simA, simB, argA, argB

    @staticmethod
    def backward(ctx, gA, gB, _ga, _gb):
        hA, hB, hnA, mwA, hnB_all, mwB_all, t, simA, simB, argA, argB = ctx.saved_tensors
        normalise, TA, TB, group, world, rank = ctx.cfg
        K = ctx.K
        if _collective(world):
            gA, gB = gA.contiguous(), gB.contiguous()
            # every rank holds a partial gradient for every B sequence: the B side runs first, and
            # its reduce-scatter (sum, each rank gets its own sequences) beside the A side's kernels
            # (the two one-sided calls share a workspace of their own: the second finds the first one's marks of
            # mutual choices in it instead of computing them again)
            ws = K.filip_bwd_workspace(hnA.shape[0], TA, hnB_all.shape[0], TB, hnA.device)
            _, dnB_all, _ = K.filip_sim_bwd(hnA, mwA, TA, hnB_all, mwB_all, TB, t, simA, simB, argA, argB, gA, gB, sides="B", ws=ws)
            dnB = torch.empty((hB.shape[0],) + tuple(dnB_all.shape[1:]), dtype=dnB_all.dtype, device=dnB_all.device)
            work = dist.reduce_scatter_tensor(dnB, dnB_all, group=group, async_op=True)
            # dt stays this rank's share, like every shared-parameter gradient: the trainer's
            # gradient all-reduce (SUM) completes it
            dnA, _, dt = K.filip_sim_bwd(hnA, mwA, TA, hnB_all, mwB_all, TB, t, simA, simB, argA, argB, gA, gB, sides="A", resume=True, ws=ws)
            dhA = K.filip_prep_bwd(hA, dnA, normalise, tsign=t)
            work.wait()
            dhB = K.filip_prep_bwd(hB, dnB, normalise)
        else:   # no exchange between the score backward and the prep backward: one fused call
            dhA, dhB, dt = K.filip_bwd_fused(hnA, mwA, hnB_all, mwB_all, t, simA, simB, argA, argB, gA.contiguous(), gB.contiguous(), hA,
                                             hB, normalise)
        return dhA, dhB, None, None, dt.reshape(t.shape).to(t.dtype), None, None, None


def filip_similarity(hA, hB, maskA, maskB, temperature, normalise: bool = False, group=None, kernels=None):
    """Token-wise max-similarity scores ``(sim_A, sim_B)``, each ``[B_own, B_global]``.

    ``hA`` are this rank's A sequences, ``hB`` this rank's B sequences (gathered over the
    group inside).  With ``normalise=True`` the per-token L2 normalisation of
    ``ExtendedFILIP.forward`` (no epsilon) is fused into the operand preparation."""
    if hA.dim() != 3 or hB.dim() != 3 or hA.shape[2] != hB.shape[2]:
        raise ValueError("filip_similarity: expected hA [B,TA,d] and hB [B,TB,d] with one d")
    if tuple(maskA.shape) != tuple(hA.shape[:2]) or tuple(maskB.shape) != tuple(hB.shape[:2]):
        raise ValueError("filip_similarity: masks must be [B,T] matching the embeddings")
    t = temperature if isinstance(temperature, torch.Tensor) else torch.tensor(float(temperature), device=hA.device)
    world, _ = _world(group)
    if world > 1:
        # the B sequences of all ranks are gathered into one tensor: ranks whose batches were padded
        # to different lengths first agree on the longest (padding tokens are masked out)
        longest = torch.tensor([hB.shape[1]], dtype=torch.int64, device=hB.device)
        dist.all_reduce(longest, op=dist.ReduceOp.MAX, group=group)
        extra = int(longest.item()) - hB.shape[1]
        if extra > 0:
            hB = torch.nn.functional.pad(hB, (0, 0, 0, extra))
            maskB = torch.nn.functional.pad(maskB, (0, extra))
    simA, simB, _, _ = _FilipSim.apply(_f32(hA), _f32(hB), maskA, maskB, _f32(t), bool(normalise), group, kernels or _hip_kernels)
    if world > 1:   # this rank's row block of the global score matrices: row_ce reads the tag
        _, rank = _world(group)
        for sim in (simA, simB):
            setattr(sim, _ROW_SHARD_TAG, (rank * hA.shape[0], world * hA.shape[0], tuple(sim.shape)))
    return simA, simB
