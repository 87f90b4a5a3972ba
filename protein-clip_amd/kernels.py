"""Tensor-level wrappers over the C ABI (include/pclip.h): argument checks, output and
workspace allocation through torch, one C call each.  No arithmetic happens here."""
from __future__ import annotations

import collections
import ctypes
import threading
from typing import Optional, Tuple

import torch

from . import _lib
from ._lib import check, ptr, require_gpu_tensor, stream_of

F32 = torch.float32
_ws_cache = collections.OrderedDict()
_WS_CACHE_STREAMS = 8   # scratch buffers kept: the streams used most recently (a program that makes streams for ever does not leak)


def _workspace(device, nbytes: int) -> torch.Tensor:
    """Scratch for one call: one buffer per (device, stream) that only ever grows.  Calls on one
    stream run in order, so they can share it; calls on different streams get different buffers.
    A buffer that is outgrown or dropped may still be in use by kernels already enqueued on its
    stream: the caching allocator is told so (``record_stream``) before the reference goes.
    Nothing may rely on the CONTENTS of this buffer from one call to the next (a caller that needs
    that — the two one-sided FILIP backward calls — brings a workspace of its own)."""
    stream = torch.cuda.current_stream(device)
    key = (device.index, stream.cuda_stream)
    buf = _ws_cache.get(key)
    if buf is None or buf.numel() < nbytes:
        if buf is not None:
            buf.record_stream(stream)
        buf = _empty((max(nbytes, 1 << 20),), torch.uint8, device)
        _ws_cache[key] = buf
    _ws_cache.move_to_end(key)
    while len(_ws_cache) > _WS_CACHE_STREAMS:
        _, old = _ws_cache.popitem(last=False)
        old.record_stream(stream)   # its own stream's kernels are long enqueued; the allocator waits for this one's as well
    return buf


class BufferArena:
    """Replays the device allocations of a call sequence that repeats: inside ``with arena:`` every output
    and scratch tensor the wrappers of this module would allocate is taken from the arena instead — created
    on the first pass, handed out again, in the same order, on every later one.  A step that runs the same
    calls on the same shapes every time (the sharded training steps) thus allocates nothing after its first
    pass.  The tensors a pass returns are overwritten by the next pass."""

    _tls = threading.local()

    def __init__(self):
        self.bufs, self.pos = [], 0

    def __enter__(self):
        self.prev = getattr(BufferArena._tls, "cur", None)
        BufferArena._tls.cur = self
        self.pos = 0
        return self

    def __exit__(self, *exc):
        BufferArena._tls.cur = self.prev
        return False

    def take(self, shape, dtype, device) -> torch.Tensor:
        shape = tuple(int(v) for v in shape)
        if self.pos < len(self.bufs):
            t = self.bufs[self.pos]
            if tuple(t.shape) != shape or t.dtype != dtype or t.device != torch.device(device):
                t = self.bufs[self.pos] = torch.empty(shape, dtype=dtype, device=device)   # the sequence changed: start over from here
                del self.bufs[self.pos + 1:]
        else:
            t = torch.empty(shape, dtype=dtype, device=device)
            self.bufs.append(t)
        self.pos += 1
        return t


def _empty(shape, dtype, device) -> torch.Tensor:
    arena = getattr(BufferArena._tls, "cur", None)
    if arena is not None:
        return arena.take(shape, dtype, device)
    return torch.empty(tuple(shape), dtype=dtype, device=device)


def _zeros(shape, dtype, device) -> torch.Tensor:
    return _empty(shape, dtype, device).zero_()


def _scalar(t, device, name):
    """Temperature / upstream-gradient scalars live on the device: accept a 0-d or 1-element
    GPU tensor, or a Python float (copied once)."""
    if isinstance(t, torch.Tensor):
        if t.numel() != 1:
            raise _lib.PclipError(f"{name}: expected a scalar tensor")
        t = t.detach().reshape(1)
        if not t.is_cuda or t.dtype != F32:
            t = t.to(device=device, dtype=F32)
        return t.contiguous()
    return torch.tensor([float(t)], dtype=F32, device=device)


# ---- masked mean over tokens ------------------------------------------------------------
def _mask_bytes(mask: torch.Tensor, shape) -> torch.Tensor:
    """[B,T] mask as one byte per token holding its VALUE (attention masks are 0/1)."""
    if mask.dtype == torch.uint8:
        m8 = mask.contiguous()
    elif mask.dtype == torch.bool:
        m8 = mask.contiguous().view(torch.uint8)
    else:
        m8 = mask.to(torch.uint8).contiguous()
    if tuple(m8.shape) != tuple(shape) or not m8.is_cuda:
        raise _lib.PclipError("masked_mean: mask must be a GPU tensor of shape [B,T]")
    return m8


def masked_mean_fwd(h: torch.Tensor, mask: torch.Tensor):
    """h [B,T,d] f32, mask [B,T] -> (mean [B,d], inv_count [B], mask bytes)."""
    require_gpu_tensor(h, F32, "h")
    B, T, d = h.shape
    m8 = _mask_bytes(mask, (B, T))
    mean = torch.empty(B, d, dtype=F32, device=h.device)
    inv = torch.empty(B, dtype=F32, device=h.device)
    check(_lib.load().pclip_masked_mean_fwd(ptr(h), ptr(m8), B, T, d, ptr(mean), ptr(inv), stream_of(h)), "pclip_masked_mean_fwd")
    return mean, inv, m8


def masked_mean_bwd(g: torch.Tensor, m8: torch.Tensor, inv: torch.Tensor, T: int) -> torch.Tensor:
    require_gpu_tensor(g, F32, "g")
    B, d = g.shape
    dh = torch.empty(B, T, d, dtype=F32, device=g.device)
    check(_lib.load().pclip_masked_mean_bwd(ptr(g), ptr(m8), ptr(inv), B, T, d, ptr(dh), stream_of(g)), "pclip_masked_mean_bwd")
    return dh


# ---- embedding head after the pooling -----------------------------------------------------
HEAD_MAX_BLOCKS = 4
HEAD_MAX_D = 128


def _head_pointers(tensors):
    """Host array of device pointers, in the order the library expects."""
    return (ctypes.c_void_p * len(tensors))(*[t.data_ptr() for t in tensors])


def _head_check(pooled, params, keep):
    require_gpu_tensor(pooled, F32, "pooled")
    B, d = pooled.shape
    if len(params) < 2 or (len(params) - 2) % 4:
        raise _lib.PclipError("head_tail: expected 4 tensors per block (W, b, gamma, beta) and the last W, b")
    nblk = (len(params) - 2) // 4
    for i, prm in enumerate(params):
        require_gpu_tensor(prm, F32, f"params[{i}]")
        want = (d, d) if (i % 4 == 0) else (d,)
        if tuple(prm.shape) != want:
            raise _lib.PclipError(f"head_tail: params[{i}] has shape {tuple(prm.shape)}, expected {want}")
    if keep is not None:
        require_gpu_tensor(keep, torch.uint8, "keep")
        if tuple(keep.shape) != (nblk, B, d):
            raise _lib.PclipError(f"head_tail: keep has shape {tuple(keep.shape)}, expected {(nblk, B, d)}")
    return B, d, nblk


def head_tail_fwd(pooled, params, keep, keep_scale: float, ln_eps: float, t):
    """pooled [B,d]; params = [W, b, gamma, beta] per block + [W, b]; keep [nblk,B,d] u8 or None.
    Returns (out [B,d], save) with ``save`` the activations the backward needs."""
    B, d, nblk = _head_check(pooled, params, keep)
    lib = _lib.load()
    tt = _scalar(t, pooled.device, "t")
    save = torch.empty(lib.pclip_head_tail_save_floats(B, d, nblk), dtype=F32, device=pooled.device)
    out = torch.empty_like(pooled)
    check(lib.pclip_head_tail_fwd(ptr(pooled), _head_pointers(params), nblk, ptr(keep), float(keep_scale), float(ln_eps), ptr(tt), B, d,
                                  ptr(save), ptr(out), stream_of(pooled)), "pclip_head_tail_fwd")
    return out, save


def head_tail_bwd(pooled, params, keep, keep_scale: float, ln_eps: float, t, save, dout, dt: Optional[torch.Tensor] = None):
    """Returns (dpooled, [gradient of each tensor of ``params``], dt); a given ``dt`` is accumulated into."""
    B, d, nblk = _head_check(pooled, params, keep)
    require_gpu_tensor(dout, F32, "dout")
    require_gpu_tensor(save, F32, "save")
    lib = _lib.load()
    tt = _scalar(t, pooled.device, "t")
    dpooled = torch.empty_like(pooled)
    dparams = [torch.empty_like(prm) for prm in params]
    if dt is None:
        dt = torch.zeros(1, dtype=F32, device=pooled.device)
    ws = _workspace(pooled.device, lib.pclip_head_tail_bwd_workspace_bytes(B, d, nblk))
    check(lib.pclip_head_tail_bwd(ptr(pooled), _head_pointers(params), nblk, ptr(keep), float(keep_scale), ptr(tt), ptr(save), ptr(dout), B, d,
                                  ptr(dpooled), _head_pointers(dparams), ptr(dt), ptr(ws), ws.numel(), stream_of(pooled)), "pclip_head_tail_bwd")
    return dpooled, dparams, dt


# ---- L2 normalise + scale ---------------------------------------------------------------
def l2norm_scale_fwd(e: torch.Tensor, t, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    require_gpu_tensor(e, F32, "e")
    B, d = e.shape
    tt = _scalar(t, e.device, "t")
    x = _empty(e.shape, F32, e.device) if out is None else out
    if out is not None:
        require_gpu_tensor(out, F32, "out")
        if out.shape != e.shape:
            raise _lib.PclipError("l2norm_scale_fwd: `out` must have the shape of `e`")
    check(_lib.load().pclip_l2norm_scale_fwd(ptr(e), ptr(tt), ptr(x), B, d, stream_of(e)), "pclip_l2norm_scale_fwd")
    return x


def l2norm_scale_bwd(e, t, dx, g=None, dt: Optional[torch.Tensor] = None):
    """Returns (de, dt); if ``dt`` (1-element f32 GPU tensor) is given it is accumulated into."""
    require_gpu_tensor(e, F32, "e")
    require_gpu_tensor(dx, F32, "dx")
    B, d = e.shape
    tt = _scalar(t, e.device, "t")
    gg = None if g is None else _scalar(g, e.device, "g")
    de = _empty(e.shape, F32, e.device)
    if dt is None:
        dt = _zeros((1,), F32, e.device)
    check(
        _lib.load().pclip_l2norm_scale_bwd(ptr(e), ptr(tt), ptr(dx), 1, 0, ptr(gg), ptr(de), ptr(dt), B, d, stream_of(e)),
        "pclip_l2norm_scale_bwd",
    )
    return de, dt


# ---- CLIP -------------------------------------------------------------------------------
def clip_fwd(X_own, Y_own, X_all, Y_all, row_offset: int = 0, logits_mode: int = 0):
    """Returns (lse_row[Br], lse_col[Br], loss_part[1], logits or None)."""
    for n, v in (("X_own", X_own), ("Y_own", Y_own), ("X_all", X_all), ("Y_all", Y_all)):
        require_gpu_tensor(v, F32, n)
    Br, d = X_own.shape
    Bg = X_all.shape[0]
    if Y_own.shape != X_own.shape or Y_all.shape != X_all.shape or X_all.shape[1] != d:
        raise _lib.PclipError("clip_fwd: shape mismatch between the two sides")
    lib = _lib.load()
    dev = X_own.device
    nbytes = lib.pclip_clip_workspace_bytes(Br, Bg, d)
    ws = _workspace(dev, nbytes)
    lse_row = _empty((Br,), F32, dev)
    lse_col = _empty((Br,), F32, dev)
    loss = _empty((1,), F32, dev)
    logits = _empty((Br, Bg), F32, dev) if logits_mode else None
    check(
        lib.pclip_clip_fwd(ptr(X_own), ptr(Y_own), ptr(X_all), ptr(Y_all), Br, Bg, d, row_offset, ptr(lse_row), ptr(lse_col),
                           ptr(loss), ptr(logits), logits_mode, ptr(ws), ws.numel(), stream_of(X_own)),
        "pclip_clip_fwd",
    )
    return lse_row, lse_col, loss, logits


def clip_bwd(X_own, Y_own, X_all, Y_all, lse_row_all, lse_col_all, row_offset: int = 0, g=None):
    """Returns (dX_own, dY_own)."""
    Br, d = X_own.shape
    Bg = X_all.shape[0]
    require_gpu_tensor(lse_row_all, F32, "lse_row_all")
    require_gpu_tensor(lse_col_all, F32, "lse_col_all")
    if lse_row_all.numel() != Bg or lse_col_all.numel() != Bg:
        raise _lib.PclipError("clip_bwd: lse vectors must cover the global batch")
    lib = _lib.load()
    dev = X_own.device
    ws = _workspace(dev, lib.pclip_clip_workspace_bytes(Br, Bg, d))
    gg = None if g is None else _scalar(g, dev, "g")
    dX = _empty(X_own.shape, F32, dev)
    dY = _empty(Y_own.shape, F32, dev)
    check(
        lib.pclip_clip_bwd(ptr(X_own), ptr(Y_own), ptr(X_all), ptr(Y_all), ptr(lse_row_all), ptr(lse_col_all), Br, Bg, d, row_offset,
                           ptr(gg), ptr(dX), ptr(dY), ptr(ws), ws.numel(), stream_of(X_own)),
        "pclip_clip_bwd",
    )
    return dX, dY


def clip_loss_grad(x, y):
    """x, y [B,d] as the towers return them -> (loss [1], d loss / d x, d loss / d y): the loss of
    training_utils.py:85-89 and its backward in one call (one launch up to B = 512)."""
    require_gpu_tensor(x, F32, "x")
    require_gpu_tensor(y, F32, "y")
    if x.shape != y.shape or x.dim() != 2:
        raise _lib.PclipError("clip_loss_grad: expected two [B,d] tensors of one shape")
    B, d = x.shape
    lib = _lib.load()
    ws = _workspace(x.device, lib.pclip_clip_loss_grad_workspace_bytes(B, d))
    loss = torch.empty(1, dtype=F32, device=x.device)
    dx, dy = torch.empty_like(x), torch.empty_like(y)
    check(lib.pclip_clip_loss_grad(ptr(x), ptr(y), B, d, ptr(loss), ptr(dx), ptr(dy), ptr(ws), ws.numel(), stream_of(x)), "pclip_clip_loss_grad")
    return loss, dx, dy


class ClipStep:
    """Pre-allocated fused step (normalise both towers -> loss -> all gradients) for one
    (B, d): every buffer is created once, ``run`` only enqueues kernels."""

    def __init__(self, B: int, d: int, device):
        lib = _lib.load()
        self.B, self.d, self.device = B, d, torch.device(device)
        self.ws = torch.empty(lib.pclip_clip_step_workspace_bytes(B, d), dtype=torch.uint8, device=device)
        self.loss = torch.empty(1, dtype=F32, device=device)
        self.de_pep = torch.empty(B, d, dtype=F32, device=device)
        self.de_rec = torch.empty(B, d, dtype=F32, device=device)
        self.dt = torch.empty(1, dtype=F32, device=device)

    def run(self, e_pep, e_rec, t, stream=None):
        st = ctypes.c_void_p(stream) if stream is not None else stream_of(e_pep)
        check(
            _lib.load().pclip_clip_step(ptr(e_pep), ptr(e_rec), ptr(t), self.B, self.d, ptr(self.loss), ptr(self.de_pep),
                                        ptr(self.de_rec), ptr(self.dt), ptr(self.ws), self.ws.numel(), st),
            "pclip_clip_step",
        )
        return self.loss, self.de_pep, self.de_rec, self.dt


def clip_step(e_pep, e_rec, t):
    require_gpu_tensor(e_pep, F32, "e_pep")
    require_gpu_tensor(e_rec, F32, "e_rec")
    B, d = e_pep.shape
    step = ClipStep(B, d, e_pep.device)
    return step.run(e_pep, e_rec, _scalar(t, e_pep.device, "t"))


# ---- row-wise InfoNCE -------------------------------------------------------------------
def rowce_fwd(logits, row_offset: int = 0, Bg: Optional[int] = None, use_dcl: bool = False):
    require_gpu_tensor(logits, F32, "logits")
    Br, Bc = logits.shape
    Bg = Br if Bg is None else Bg
    lse = torch.empty(Br, dtype=F32, device=logits.device)
    loss = torch.empty(1, dtype=F32, device=logits.device)
    check(_lib.load().pclip_rowce_fwd(ptr(logits), Br, Bc, row_offset, Bg, int(use_dcl), ptr(lse), ptr(loss), stream_of(logits)),
          "pclip_rowce_fwd")
    return lse, loss


def rowce_bwd(logits, lse, row_offset: int = 0, Bg: Optional[int] = None, use_dcl: bool = False, g=None, gscale: float = 1.0):
    Br, Bc = logits.shape
    Bg = Br if Bg is None else Bg
    gg = None if g is None else _scalar(g, logits.device, "g")
    out = torch.empty_like(logits)
    check(_lib.load().pclip_rowce_bwd(ptr(logits), ptr(lse), Br, Bc, row_offset, Bg, int(use_dcl), ptr(gg), gscale, ptr(out),
                                      stream_of(logits)), "pclip_rowce_bwd")
    return out


def filip_loss(simA, simB, row_offset: int = 0, Bg: Optional[int] = None):
    """Both row InfoNCE losses of the FILIP scores and their gradients (training_utils_filip.py:26-39).
    Returns (loss3, gA, gB): loss3 = [(lossA + lossB) / 2, lossA, lossB] for these rows' share."""
    require_gpu_tensor(simA, F32, "simA")
    require_gpu_tensor(simB, F32, "simB")
    Br, Bc = simA.shape
    Bg = Br if Bg is None else Bg
    dev = simA.device
    lse2 = _empty((2, Br), F32, dev)
    terms = _empty((2, Br), F32, dev)
    loss3 = _empty((3,), F32, dev)
    gA, gB = _empty(simA.shape, F32, dev), _empty(simB.shape, F32, dev)
    check(_lib.load().pclip_filip_loss(ptr(simA), ptr(simB), Br, Bc, row_offset, Bg, ptr(lse2), ptr(loss3), ptr(gA), ptr(gB), ptr(terms),
                                       stream_of(simA)), "pclip_filip_loss")
    return loss3, gA, gB


# ---- FILIP ------------------------------------------------------------------------------
def _tp(T: int) -> int:
    return (T + 31) // 32 * 32


def filip_prep(h, mask, normalise: bool, tsign=None, out=None):
    """h [B,T,d] f32, mask [B,T] (any integer/bool dtype) -> (hn [B,Tp,128] bf16 bits as int16,
    maskw [B,Tp/32] int32).  ``tsign``: optional device scalar; the image is negated when it
    is negative (pass the temperature on the A side).  ``out``: optional (hn, maskw) to write into
    (a rank's slice of the gathered buffers, so that nothing is copied in front of the all-gather)."""
    require_gpu_tensor(h, F32, "h")
    B, T, d = h.shape
    # the kernel reads one byte per token and treats any nonzero byte as valid (as .bool() does)
    if mask.dtype == torch.uint8:
        m8 = mask.contiguous()
    elif mask.dtype == torch.bool:
        m8 = mask.contiguous().view(torch.uint8)
    else:
        m8 = (mask != 0).to(torch.uint8).contiguous()
    if m8.shape != (B, T) or not m8.is_cuda:
        raise _lib.PclipError("filip_prep: mask must be a GPU tensor of shape [B,T]")
    Tp = _tp(T)
    hn = _empty((B, Tp, 128), torch.int16, h.device) if out is None else out[0]
    mw = _empty((B, Tp // 32), torch.int32, h.device) if out is None else out[1]
    if tuple(hn.shape) != (B, Tp, 128) or tuple(mw.shape) != (B, Tp // 32) or hn.dtype != torch.int16 or mw.dtype != torch.int32 \
            or not (hn.is_contiguous() and mw.is_contiguous()):
        raise _lib.PclipError("filip_prep: `out` must be contiguous (int16 [B,Tp,128], int32 [B,Tp/32]) tensors")
    ts = None if tsign is None else _scalar(tsign, h.device, "tsign")
    check(_lib.load().pclip_filip_prep(ptr(h), ptr(m8), B, T, d, int(normalise), ptr(ts), ptr(hn), ptr(mw), stream_of(h)),
          "pclip_filip_prep")
    return hn, mw


def filip_sim_fwd(hnA, mwA, TA: int, hnB, mwB, TB: int, t, want_arg: bool = True, unit_norm: bool = False, stream_wgs: int = 0,
                  out=None, col_offset: int = 0):
    """``stream_wgs`` (tests, tuning): number of workgroups the unit-norm kernel aims at per launch
    (0 = the library's default); the results do not depend on it.
    ``out`` = (simA, simB, argA, argB) of a WIDER problem ([Br,Bc_total] and [Br,Bc_total,Tp]) together with
    ``col_offset``: this call fills their columns col_offset .. col_offset + Bc - 1 and returns them."""
    dev = hnA.device
    Br, Bc = hnA.shape[0], hnB.shape[0]
    tt = _scalar(t, dev, "t")
    if out is None:
        simA = _empty((Br, Bc), F32, dev)
        simB = _empty((Br, Bc), F32, dev)
        argA = _empty((Br, Bc, _tp(TA)), torch.int16, dev) if want_arg else None
        argB = _empty((Br, Bc, _tp(TB)), torch.int16, dev) if want_arg else None
        total = Bc
    else:
        simA, simB, argA, argB = out
        total = simA.shape[1]
        if tuple(simA.shape) != (Br, total) or tuple(simB.shape) != (Br, total) or (argA is not None and tuple(argA.shape) != (Br, total, _tp(TA))) \
                or (argB is not None and tuple(argB.shape) != (Br, total, _tp(TB))) or not (0 <= col_offset and col_offset + Bc <= total):
            raise _lib.PclipError("filip_sim_fwd: `out` does not match a [Br, Bc_total] problem that holds these columns")
    lib = _lib.load()
    un = (max(int(stream_wgs), 2) if stream_wgs else 1) if unit_norm else 0
    nws = lib.pclip_filip_sim_fwd_workspace_bytes(Br, TA, Bc, TB, un)   # sequences beyond 256 tokens, unit norm: segments on the fast kernel
    ws = _workspace(dev, nws) if nws else None
    check(lib.pclip_filip_sim_fwd_ws(ptr(hnA), ptr(mwA), Br, TA, ptr(hnB), ptr(mwB), Bc, TB, ptr(tt), un, ptr(simA), ptr(simB), ptr(argA), ptr(argB),
                                     int(col_offset), int(total), ptr(ws), ws.numel() if ws is not None else 0, stream_of(hnA)),
          "pclip_filip_sim_fwd_ws")
    return simA, simB, argA, argB


def filip_bwd_workspace(Br: int, TA: int, Bc: int, TB: int, device) -> torch.Tensor:
    """A workspace of the caller's own for ``filip_sim_bwd(..., ws=...)``: needed when a second, one-sided call
    (``resume=True``) is to find what the first one left in it."""
    return _empty((_lib.load().pclip_filip_bwd_workspace_bytes(Br, TA, Bc, TB),), torch.uint8, device)


def filip_sim_bwd(hnA, mwA, TA, hnB, mwB, TB, t, simA, simB, argA, argB, gA, gB, dt: Optional[torch.Tensor] = None, sides: str = "AB",
                  resume: bool = False, ws: Optional[torch.Tensor] = None):
    """Returns (dnA, dnB, dt).  ``sides``: "AB" (both), "A" (dnA and dt; dnB is None) or "B" (dnB only;
    dnA and dt are None) — the multi-GPU step runs "B" first and reduce-scatters dnB while "A" runs.
    ``resume``: this is the second of two one-sided calls with the same inputs and the same ``ws``
    (from :func:`filip_bwd_workspace`), which still holds the first call's marks of mutual choices."""
    dev = hnA.device
    Br, Bc = hnA.shape[0], hnB.shape[0]
    tt = _scalar(t, dev, "t")
    require_gpu_tensor(gA, F32, "gA")
    require_gpu_tensor(gB, F32, "gB")
    dnA = _empty((Br, _tp(TA), 128), F32, dev) if "A" in sides else None
    dnB = _empty((Bc, _tp(TB), 128), F32, dev) if "B" in sides else None
    if dt is None:
        dt = _zeros((1,), F32, dev)
    lib = _lib.load()
    if resume and ws is None:
        raise _lib.PclipError("filip_sim_bwd: resume=True needs the workspace (ws=) the first one-sided call ran on")
    if ws is None:
        ws = _workspace(dev, lib.pclip_filip_bwd_workspace_bytes(Br, TA, Bc, TB))
    elif ws.numel() < lib.pclip_filip_bwd_workspace_bytes(Br, TA, Bc, TB) or ws.dtype != torch.uint8 or not ws.is_cuda:
        raise _lib.PclipError("filip_sim_bwd: ws is too small for this problem (see filip_bwd_workspace)")
    entry = lib.pclip_filip_sim_bwd_resume if resume else lib.pclip_filip_sim_bwd
    check(entry(ptr(hnA), ptr(mwA), Br, TA, ptr(hnB), ptr(mwB), Bc, TB, ptr(tt), ptr(simA), ptr(simB),
                ptr(argA), ptr(argB), ptr(gA), ptr(gB), ptr(dnA), ptr(dnB), ptr(dt), ptr(ws), ws.numel(),
                stream_of(hnA)), "pclip_filip_sim_bwd")
    return dnA, dnB, (dt if "A" in sides else None)


def filip_bwd_fused(hnA, mwA, hnB, mwB, t, simA, simB, argA, argB, gA, gB, hA, hB, normalise: bool,
                    dt: Optional[torch.Tensor] = None):
    """Score backward and prep backward in one call (single-GPU path: no gradient exchange between
    the two).  hA [Br,TA,d], hB [Bc,TB,d]: the raw embeddings given to ``filip_prep`` (A side with
    ``tsign=t``).  Returns (dhA, dhB, dt)."""
    dev = hnA.device
    for n, v in (("gA", gA), ("gB", gB), ("hA", hA), ("hB", hB)):
        require_gpu_tensor(v, F32, n)
    (Br, TA, d), (Bc, TB, _) = hA.shape, hB.shape
    tt = _scalar(t, dev, "t")
    dhA, dhB = _empty(hA.shape, F32, dev), _empty(hB.shape, F32, dev)
    if dt is None:
        dt = _zeros((1,), F32, dev)
    lib = _lib.load()
    ws = _workspace(dev, lib.pclip_filip_bwd_fused_workspace_bytes(Br, TA, Bc, TB))
    check(lib.pclip_filip_bwd_fused(ptr(hnA), ptr(mwA), Br, TA, ptr(hnB), ptr(mwB), Bc, TB, ptr(tt), ptr(simA), ptr(simB), ptr(argA),
                                    ptr(argB), ptr(gA), ptr(gB), ptr(hA), ptr(hB), d, int(normalise), ptr(dhA), ptr(dhB), ptr(dt), ptr(ws),
                                    ws.numel(), stream_of(hnA)), "pclip_filip_bwd_fused")
    return dhA, dhB, dt


def filip_prep_bwd(h, dn, normalise: bool, tsign=None, out: Optional[torch.Tensor] = None):
    require_gpu_tensor(h, F32, "h")
    require_gpu_tensor(dn, F32, "dn")
    B, T, d = h.shape
    acc = out is not None
    dh = out if acc else _empty(h.shape, F32, h.device)
    ts = None if tsign is None else _scalar(tsign, h.device, "tsign")
    check(_lib.load().pclip_filip_prep_bwd(ptr(h), ptr(dn), B, T, d, int(normalise), ptr(ts), int(acc), ptr(dh), stream_of(h)),
          "pclip_filip_prep_bwd")
    return dh


class FilipStep:
    """Pre-allocated FILIP hot-path step for one shape: raw token embeddings -> normalise ->
    scores -> both row InfoNCE losses -> every gradient.  ``run`` only enqueues kernels on the
    current stream (10 C calls); ``sim_events`` optionally brackets the score kernel with the
    two given ``torch.cuda.Event``s so its duration can be read after a sync."""

    def __init__(self, Br: int, Bc: int, TA: int, TB: int, d: int, device, row_offset: int = 0, Bg=None):
        dev = torch.device(device)
        self.shape = (Br, Bc, TA, TB, d)
        self.row_offset, self.Bg = row_offset, (Bc if Bg is None else Bg)
        TAp, TBp = _tp(TA), _tp(TB)
        e = lambda *s, dt=F32: torch.empty(*s, dtype=dt, device=dev)
        self.hnA, self.mwA = e(Br, TAp, 128, dt=torch.int16), e(Br, TAp // 32, dt=torch.int32)
        self.hnB, self.mwB = e(Bc, TBp, 128, dt=torch.int16), e(Bc, TBp // 32, dt=torch.int32)
        self.simA, self.simB = e(Br, Bc), e(Br, Bc)
        self.argA, self.argB = e(Br, Bc, TAp, dt=torch.int16), e(Br, Bc, TBp, dt=torch.int16)
        self.lse2, self.terms, self.loss3 = e(2, Br), e(2, Br), e(3)
        self.gA, self.gB = e(Br, Bc), e(Br, Bc)
        self.dhA, self.dhB = e(Br, TA, d), e(Bc, TB, d)
        self.dt = e(1)
        self.bws = e(_lib.load().pclip_filip_bwd_fused_workspace_bytes(Br, TA, Bc, TB), dt=torch.uint8)

    def run(self, hA, hB, mA8, mB8, t, sim_events=None, bwd_events=None):
        """hA [Br,TA,d], hB [Bc,TB,d] f32; mA8/mB8 uint8 masks; t 1-element f32 tensor (all on
        the GPU).  Returns (lossA, lossB, dhA, dhB, dt); the loss is (lossA + lossB) / 2.
        ``bwd_events`` brackets the backward call (marks, inversions, gathers, prep backward)."""
        lib = _lib.load()
        Br, Bc, TA, TB, d = self.shape
        st = stream_of(hA)
        off, Bg = self.row_offset, self.Bg
        check(lib.pclip_filip_prep(ptr(hA), ptr(mA8), Br, TA, d, 1, ptr(t), ptr(self.hnA), ptr(self.mwA), st), "pclip_filip_prep")
        check(lib.pclip_filip_prep(ptr(hB), ptr(mB8), Bc, TB, d, 1, None, ptr(self.hnB), ptr(self.mwB), st), "pclip_filip_prep")
        if sim_events is not None:
            sim_events[0].record()
        check(lib.pclip_filip_sim_fwd(ptr(self.hnA), ptr(self.mwA), Br, TA, ptr(self.hnB), ptr(self.mwB), Bc, TB, ptr(t), 1, ptr(self.simA),
                                      ptr(self.simB), ptr(self.argA), ptr(self.argB), st), "pclip_filip_sim_fwd")
        if sim_events is not None:
            sim_events[1].record()
        check(lib.pclip_filip_loss(ptr(self.simA), ptr(self.simB), Br, Bc, off, Bg, ptr(self.lse2), ptr(self.loss3), ptr(self.gA), ptr(self.gB),
                                   ptr(self.terms), st), "pclip_filip_loss")
        self.dt.zero_()
        if bwd_events is not None:
            bwd_events[0].record()
        check(lib.pclip_filip_bwd_fused(ptr(self.hnA), ptr(self.mwA), Br, TA, ptr(self.hnB), ptr(self.mwB), Bc, TB, ptr(t), ptr(self.simA),
                                        ptr(self.simB), ptr(self.argA), ptr(self.argB), ptr(self.gA), ptr(self.gB), ptr(hA), ptr(hB), d, 1,
                                        ptr(self.dhA), ptr(self.dhB), ptr(self.dt), ptr(self.bws), self.bws.numel(), st),
              "pclip_filip_bwd_fused")
        if bwd_events is not None:
            bwd_events[1].record()
        return self.loss3[1:2], self.loss3[2:3], self.dhA, self.dhB, self.dt
