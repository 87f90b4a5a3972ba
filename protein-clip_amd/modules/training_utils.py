"""CLIP train / eval loops and loss functions with the reference's names and signatures
(modules/training_utils.py).

``_contrastive_loss`` / ``_contrastive_loss_gc`` / ``get_logits`` run on the HIP kernels.
``train_gc`` keeps GradCache's semantics — no-grad forward of every sub-batch, ONE loss over
the concatenated representations, re-forward with the cached representation gradients — but
needs no ``grad_cache`` package, and when ``torch.distributed`` is initialised the
concatenation additionally spans every rank through an RCCL all-gather inside the loss
(each rank then back-propagates its own rows only; head gradients are summed over ranks).
"""
from __future__ import annotations

import torch
import torch.distributed as dist

from .. import ops


# --------------------------------------------------------------------------------------
# losses
# --------------------------------------------------------------------------------------
def _contrastive_loss(pep_embedding: torch.Tensor, rec_embedding: torch.Tensor) -> torch.Tensor:
    """Symmetric InfoNCE of ``pep_embedding [B,d] @ rec_embedding [d,B]``
    (training_utils.py:74-89; the caller passes ``rec.t()``)."""
    return ops.clip_loss(pep_embedding, rec_embedding.t())


def _cat(x):
    return torch.cat(list(x), dim=0) if isinstance(x, (list, tuple)) else x


def _contrastive_loss_gc(x, y, group=None) -> torch.Tensor:
    """Same loss on lists of sub-batch representations (training_utils.py:247-265); with a
    process group the batch is the concatenation over all ranks."""
    return ops.clip_loss(_cat(x), _cat(y), group=group)


def get_logits(x, y) -> torch.Tensor:
    """``exp(x @ y.T)`` of the concatenated lists (training_utils.py:157-172)."""
    return ops.clip_exp_logits(_cat(x), _cat(y))


# --------------------------------------------------------------------------------------
# plain loops
# --------------------------------------------------------------------------------------
def _tokenise(tokenizer, sequences, device):
    return tokenizer(sequences, return_tensors="pt", padding=True).to(device)


def _process_batch(model, batch_data, tokenizer, device, compute_grad=False):
    peptides, receptors = batch_data
    pep, rec = model(_tokenise(tokenizer, peptides, device), _tokenise(tokenizer, receptors, device))
    loss = _contrastive_loss(pep, rec.t())
    if compute_grad:
        loss.backward()
    return loss.item()


def _distributed() -> bool:
    return dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1


def train(model, data_loader, optimizer, tokenizer, device):
    """One epoch, one optimizer step per batch; returns the mean batch loss.  Under
    ``torch.distributed`` the loss of a batch spans the batches of all ranks (all-gather inside
    the loss) and every rank back-propagates its own rows, so the head gradients are summed over
    the ranks before the step."""
    model.train()
    running = 0.0
    for batch in data_loader:
        optimizer.zero_grad()
        running += _process_batch(model, batch, tokenizer, device, compute_grad=True)
        if _distributed():
            _sum_head_grads(model)
        optimizer.step()
    return running / len(data_loader)


def evaluate(model, data_loader, tokenizer, device):
    model.eval()
    running = 0.0
    with torch.no_grad():
        for batch in data_loader:
            running += _process_batch(model, batch, tokenizer, device)
    return running / len(data_loader)


# --------------------------------------------------------------------------------------
# big-batch training (GradCache semantics, optionally across ranks)
# --------------------------------------------------------------------------------------
class _RngSnapshot:
    """Random state of the CPU and of one GPU, so a sub-batch can be re-run with the very
    dropout masks of its first pass."""

    def __init__(self, device):
        self.cpu = torch.get_rng_state()
        self.dev = torch.device(device)
        self.cuda = torch.cuda.get_rng_state(self.dev) if self.dev.type == "cuda" else None

    def restore(self):
        torch.set_rng_state(self.cpu)
        if self.cuda is not None:
            torch.cuda.set_rng_state(self.cuda, self.dev)


def _encode_cached(encoder, inputs, device):
    """First pass of one sub-batch: representation without a graph (a leaf that will receive
    the loss gradient) plus what is needed to replay the pass."""
    snap = _RngSnapshot(device)
    with torch.no_grad():
        rep = encoder(inputs)
    return rep.detach().requires_grad_(True), snap


def _replay(encoder, inputs, snap, rep):
    """Second pass with a graph under the first pass's random state; pushes the cached
    representation gradient through it (the surrogate is <rep_again, d loss / d rep>)."""
    keep = _RngSnapshot(snap.dev)
    snap.restore()
    again = encoder(inputs)
    keep.restore()
    torch.autograd.backward(again, grad_tensors=rep.grad)


def _sum_head_grads(model, group=None):
    """Every rank back-propagated the global loss through its own rows only, so the gradient
    of a shared parameter is the SUM over ranks (one flat all-reduce)."""
    grads = [p.grad for p in model.parameters() if p.requires_grad and p.grad is not None]
    if not grads:
        return
    flat = torch.cat([g.reshape(-1) for g in grads])
    dist.all_reduce(flat, group=group)
    offset = 0
    for g in grads:
        g.copy_(flat[offset:offset + g.numel()].view_as(g))
        offset += g.numel()


def train_gc(model, data_loader, tokenizer, optimizer, scaler, device, accumulated_batches=1, group=None):
    """Big-batch epoch (training_utils.py:92-155): every ``accumulated_batches`` sub-batches
    form one contrastive batch and one optimizer step.  Trailing sub-batches that do not fill
    a batch are encoded and dropped, as in the reference.  Returns the mean big-batch loss."""
    model.train()
    distributed = _distributed()
    total, big_batches = 0.0, 0
    pending = []
    for step, (xx, yy) in enumerate(data_loader):
        xx = _tokenise(tokenizer, xx, device)
        yy = _tokenise(tokenizer, yy, device)
        xx["temperature"] = model.temperature
        yy["temperature"] = model.temperature
        rx, sx = _encode_cached(model.pep_encoder, xx, device)
        ry, sy = _encode_cached(model.rec_encoder, yy, device)
        pending.append((xx, yy, rx, ry, sx, sy))
        if (step + 1) % accumulated_batches:
            continue
        big_batches += 1
        print(step)
        loss = _contrastive_loss_gc([p[2] for p in pending], [p[3] for p in pending], group=group if distributed else None)
        total += loss.item()
        scaler.scale(loss).backward()
        for xx_, yy_, rx_, ry_, sx_, sy_ in pending:
            _replay(model.pep_encoder, xx_, sx_, rx_)
            _replay(model.rec_encoder, yy_, sy_, ry_)
        pending = []
        if distributed:
            _sum_head_grads(model, group)
        scaler.step(optimizer)
        scaler.update()
        optimizer.zero_grad()
    return total / max(big_batches, 1)


def eval_gc_allrec_onepep(model, data_loader, device, tokenizer, agg_batches=2, k=0):
    """Rank of peptide ``k`` against the receptors of the first ``agg_batches`` sub-batches
    (training_utils.py:174-228): every row of the score matrix is that one peptide, so the
    diagonal holds its score against each receptor; returns ``[1 + number of receptors scoring
    higher than receptor k]`` after the first aggregated batch, like the reference."""
    model.eval()
    chosen, seen = None, 0
    for xx, yy in data_loader:
        if k // len(xx) == seen:
            chosen = xx[k % len(yy)]
        seen += 1
    reps_x, reps_y, ranks = [], [], []
    with torch.no_grad():
        for step, (xx, yy) in enumerate(data_loader):
            xx = _tokenise(tokenizer, (chosen,) * len(xx), device)
            yy = _tokenise(tokenizer, yy, device)
            xx["temperature"] = model.temperature
            yy["temperature"] = model.temperature
            reps_x.append(model.pep_encoder(xx))
            reps_y.append(model.rec_encoder(yy))
            if (step + 1) % agg_batches == 0:
                scores = torch.diag(get_logits(reps_x, reps_y))
                ranks.append(int((scores > scores[k]).sum().item()) + 1)
                return ranks
    return ranks
