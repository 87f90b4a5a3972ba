"""FILIP loss functions and loops with the reference's names and signatures
(modules/training_utils_filip.py); the row InfoNCE runs on the HIP kernels."""
from __future__ import annotations

import torch

from .. import ops
from .training_utils import _distributed, _sum_head_grads


def _contrastive_loss(logits: torch.Tensor, use_dcl: bool = False) -> torch.Tensor:
    """``mean_i(-log E_ii + log sum_j E_ij)``, ``E = exp(logits)``; ``use_dcl`` removes the
    diagonal from the denominator (training_utils_filip.py:4-23).

    Under ``torch.distributed`` ``ExtendedFILIP.forward`` returns this rank's rows against the B
    sequences of all ranks, ``[B, W*B]``: the operator recognises that shape, takes column
    ``rank*B + i`` as the target of row i and returns the mean over the global batch."""
    return ops.row_ce(logits, use_dcl=use_dcl)


def _compute_loss(sim_scores_A: torch.Tensor, sim_scores_B: torch.Tensor) -> torch.Tensor:
    """Average of the two row-wise losses (training_utils_filip.py:26-39)."""
    return (_contrastive_loss(sim_scores_A) + _contrastive_loss(sim_scores_B)) / 2


def _tokenise(tokenizer, sequences, device):
    return tokenizer(sequences, return_tensors="pt", padding=True).to(device)


def _process_batch(model, batch_data, tokenizer, device, compute_grad=False):
    peptides, receptors = batch_data
    sim_A, sim_B = model(_tokenise(tokenizer, peptides, device), _tokenise(tokenizer, receptors, device))
    loss = _compute_loss(sim_A, sim_B)
    if compute_grad:
        loss.backward()
    return loss.item()


def train(model, data_loader, optimizer, tokenier, device):
    """One epoch, one optimizer step per batch; returns the mean batch loss.  (The fourth
    parameter keeps the reference's spelling so keyword calls stay valid.)  Under
    ``torch.distributed`` every rank back-propagates the global loss through its own rows, so the
    head and temperature gradients are summed over the ranks before the step."""
    model.train()
    running = 0.0
    for batch in data_loader:
        optimizer.zero_grad()
        running += _process_batch(model, batch, tokenier, device, compute_grad=True)
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
