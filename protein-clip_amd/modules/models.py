"""Projection heads and dual encoders with the reference's API (modules/models.py).

Everything up to the head output is ordinary PyTorch-ROCm (frozen ESM trunk, Linear /
LayerNorm / ReLU / Dropout).  The lines of the reference that sit on the contrastive hot path
are replaced by HIP operators:

* ``Encoder._masked_mean`` (models.py:76-90) -> :func:`protein_clip_amd.ops.masked_mean`;
* ``Encoder.forward`` tail, ``embedding_ffn`` and ``F.normalize(e) * exp(t/2)`` (models.py:49-51)
  -> :func:`protein_clip_amd.ops.embedding_head` (one launch; the layers stay ordinary
  ``nn.Linear`` / ``nn.LayerNorm`` modules that own the parameters);
* ``ExtendedFILIP.forward`` token normalisation + ``_filip_similarity_score``
  (models.py:237-243, 292-356) -> :func:`protein_clip_amd.ops.filip_similarity`.

Sub-module names match the reference so its ``state_dict`` files load unchanged
(``pep_encoder.projection.weight``, ``pep_encoder.amino_acid_ffn.0.0.weight``, ...,
``temperature``).
"""
from __future__ import annotations

import torch
import torch.nn as nn

from .. import ops


def _feed_forward(width: int, depth: int, dropout_rate: float) -> nn.Sequential:
    """``depth - 1`` blocks of Linear -> ReLU -> LayerNorm -> Dropout, then one Linear
    (the layout behind models.py:54-73; child indices are part of the checkpoint keys)."""
    blocks = []
    for _ in range(depth - 1):
        blocks.append(nn.Sequential(nn.Linear(width, width), nn.ReLU(inplace=True), nn.LayerNorm(width), nn.Dropout(dropout_rate)))
    blocks.append(nn.Linear(width, width))
    return nn.Sequential(*blocks)


def _token_features(tower, seq):
    hidden = tower.esm_model(input_ids=seq["input_ids"], attention_mask=seq["attention_mask"]).last_hidden_state
    return tower.amino_acid_ffn(tower.projection(hidden))


def _head_layout(ffn: nn.Sequential):
    """``(params, dropout_p, ln_eps)`` when ``ffn`` has the layout ``_feed_forward`` builds (blocks of
    Linear -> ReLU -> LayerNorm -> Dropout, then one Linear, square, within the fused kernel's
    limits), else None."""
    mods = list(ffn)
    if not mods or not isinstance(mods[-1], nn.Linear) or len(mods) - 1 > ops.HEAD_MAX_BLOCKS:
        return None
    width = mods[-1].in_features
    params, p, eps = [], 0.0, 1e-5
    for blk in mods[:-1]:
        if not (isinstance(blk, nn.Sequential) and len(blk) == 4 and isinstance(blk[0], nn.Linear) and isinstance(blk[1], nn.ReLU)
                and isinstance(blk[2], nn.LayerNorm) and isinstance(blk[3], nn.Dropout)):
            return None
        if blk[0].bias is None or not blk[2].elementwise_affine or blk[2].bias is None or tuple(blk[2].normalized_shape) != (width,):
            return None
        params += [blk[0].weight, blk[0].bias, blk[2].weight, blk[2].bias]
        p, eps = blk[3].p, blk[2].eps
    if mods[-1].bias is None:
        return None
    params += [mods[-1].weight, mods[-1].bias]
    if any(tuple(w.shape) != (width, width) for w in params[0::4]) or width % 4 or width > ops.HEAD_MAX_D:
        return None
    if len({(blk[3].p, blk[2].eps) for blk in mods[:-1]}) > 1:
        return None
    return params, p, eps


class Encoder(nn.Module):
    """One CLIP tower: trunk -> projection -> per-residue FFN -> masked mean -> FFN ->
    unit-norm embedding scaled by exp(temperature / 2)."""

    def __init__(self, input_dim, embedding_dim, h1, h2, dropout_rate, esm_model):
        super().__init__()
        self.projection = nn.Linear(input_dim, embedding_dim)
        self.amino_acid_ffn = _feed_forward(embedding_dim, h1, dropout_rate)
        self.embedding_ffn = _feed_forward(embedding_dim, h2, dropout_rate)
        self.esm_model = esm_model

    def forward(self, seq):
        """``seq``: mapping with ``input_ids``, ``attention_mask`` and ``temperature``."""
        pooled = self._masked_mean(_token_features(self, seq), seq["attention_mask"])
        layout = _head_layout(self.embedding_ffn)
        if layout is None:   # a head of another shape: its own modules, then the normalise kernel
            return ops.l2norm_scale(self.embedding_ffn(pooled), seq["temperature"])
        params, p, eps = layout
        blocks = len(self.embedding_ffn) - 1
        keep, scale = None, 1.0
        if self.training and p > 0.0 and blocks > 0:
            # nn.Dropout's element-wise Bernoulli(1 - p) masks, drawn from torch's generator of this device
            keep = (torch.rand((blocks,) + tuple(pooled.shape), device=pooled.device) >= p).to(torch.uint8)
            scale = 1.0 / (1.0 - p) if p < 1.0 else 0.0
        return ops.embedding_head(pooled, params, seq["temperature"], keep, scale, eps)

    _build_ffn = staticmethod(_feed_forward)

    @staticmethod
    def _masked_mean(h, mask):
        """Masked mean over tokens (models.py:76-90) on the HIP kernel."""
        return ops.masked_mean(h.float(), mask).to(h.dtype)


class ExtendedCLIP(nn.Module):
    def __init__(self, input_dim, embedding_dim, h1, h2, dropout, esm_model):
        super().__init__()
        self.pep_encoder = Encoder(input_dim, embedding_dim, h1, h2, dropout, esm_model)
        self.rec_encoder = Encoder(input_dim, embedding_dim, h1, h2, dropout, esm_model)
        self.temperature = nn.Parameter(torch.tensor(1.0))

    def forward(self, pep_seq, rec_seq):
        """Returns ``(pep_embedding [B,d], rec_embedding [B,d])``; like the reference it writes
        the temperature into both input mappings."""
        pep_seq["temperature"] = self.temperature
        rec_seq["temperature"] = self.temperature
        return self.pep_encoder(pep_seq), self.rec_encoder(rec_seq)


class FILIPEncoder(nn.Module):
    """One FILIP tower: per-residue embeddings and the attention mask (``embedding_ffn`` is
    built but unused, as in the reference, so checkpoints keep their keys)."""

    def __init__(self, input_dim, embedding_dim, h1, h2, dropout_rate, esm_model):
        super().__init__()
        self.projection = nn.Linear(input_dim, embedding_dim)
        self.amino_acid_ffn = _feed_forward(embedding_dim, h1, dropout_rate)
        self.embedding_ffn = _feed_forward(embedding_dim, h2, dropout_rate)
        self.esm_model = esm_model

    _build_ffn = staticmethod(_feed_forward)

    def forward(self, seq):
        return _token_features(self, seq), seq["attention_mask"]


class ExtendedFILIP(nn.Module):
    def __init__(self, input_dim, embedding_dim, h1, h2, dropout, esm_model):
        super().__init__()
        self.pep_encoder = FILIPEncoder(input_dim, embedding_dim, h1, h2, dropout, esm_model)
        self.rec_encoder = FILIPEncoder(input_dim, embedding_dim, h1, h2, dropout, esm_model)
        self.temperature = nn.Parameter(torch.tensor(1.0))

    def forward(self, pep_seq, rec_seq, group=None):
        """Returns ``(sim_A, sim_B)``, each ``[B, B_global]`` (``B_global = B`` without a
        process group).  Token normalisation (no epsilon) is fused into the score operator."""
        pep_seq["temperature"] = self.temperature
        rec_seq["temperature"] = self.temperature
        pep_tokens, pep_mask = self.pep_encoder(pep_seq)
        rec_tokens, rec_mask = self.rec_encoder(rec_seq)
        return ops.filip_similarity(pep_tokens, rec_tokens, pep_mask, rec_mask, self.temperature, normalise=True, group=group)

    @staticmethod
    def _masked_mean(t, mask, dim=1, eps=1e-6):
        kept = torch.where(mask, t, torch.zeros((), dtype=t.dtype, device=t.device))
        return kept.sum(dim=dim) / mask.sum(dim=dim).clamp(min=eps)

    @staticmethod
    def _filip_similarity_score(hA, hB, maskA, maskB, temperature, include_group: bool = False):
        """Scores of already-prepared token embeddings (no normalisation here, as in
        models.py:292-356).  With ``include_group`` the inputs carry a leading group axis
        (``[m,b,t,d]`` / ``[n,b,t,d]``) and the result is ``[m,n,bA,bB]``."""
        if not include_group:
            return ops.filip_similarity(hA, hB, maskA, maskB, temperature, normalise=False)
        m, bA = hA.shape[:2]
        n, bB = hB.shape[:2]
        sA, sB = ops.filip_similarity(hA.reshape((m * bA,) + tuple(hA.shape[2:])), hB.reshape((n * bB,) + tuple(hB.shape[2:])),
                                      maskA.reshape(m * bA, -1), maskB.reshape(n * bB, -1), temperature, normalise=False)
        shape = lambda s: s.reshape(m, bA, n, bB).permute(0, 2, 1, 3)
        return shape(sA), shape(sB)
