"""Offline stand-ins for what the reference downloads: an ESM-2 tokenizer built from the
33-token ESM alphabet, a randomly initialised ESM-2 trunk of any size, and random protein
sequence pairs.  Used by the entry points in ``--synthetic`` mode, the smoke paths and the
tests; none of it touches the network."""
from __future__ import annotations

import os
import random
import tempfile

import torch
from torch.utils.data import Dataset

ESM_ALPHABET = ["<cls>", "<pad>", "<eos>", "<unk>", "L", "A", "G", "V", "S", "E", "R", "T", "I", "D", "P", "K", "Q", "N", "F",
                "Y", "M", "H", "W", "C", "X", "B", "U", "Z", "O", ".", "-", "<null_1>", "<mask>"]
RESIDUES = "ACDEFGHIKLMNPQRSTVWY"


def make_tokenizer(directory: str | None = None):
    from transformers import EsmTokenizer
    directory = directory or tempfile.mkdtemp(prefix="esm_vocab_")
    path = os.path.join(directory, "vocab.txt")
    with open(path, "w") as f:
        f.write("\n".join(ESM_ALPHABET) + "\n")
    return EsmTokenizer(vocab_file=path)


def make_esm(hidden_size: int = 640, num_layers: int = 30, num_heads: int = 20):
    """ESM-2 architecture with random weights (defaults = esm2_t30_150M_UR50D's shape)."""
    from transformers import EsmConfig, EsmModel
    cfg = EsmConfig(vocab_size=33, hidden_size=hidden_size, num_hidden_layers=num_layers, num_attention_heads=num_heads,
                    intermediate_size=4 * hidden_size, position_embedding_type="rotary", token_dropout=True, pad_token_id=1,
                    mask_token_id=32, max_position_embeddings=1026, emb_layer_norm_before=False, hidden_dropout_prob=0.0,
                    attention_probs_dropout_prob=0.0)
    return EsmModel(cfg, add_pooling_layer=False)


def random_sequences(n: int, lo: int, hi: int, seed: int):
    rng = random.Random(seed)
    return ["".join(rng.choice(RESIDUES) for _ in range(rng.randint(lo, hi))) for _ in range(n)]


class PairDataset(Dataset):
    """(peptide, receptor) string pairs, like the reference's datasets yield."""

    def __init__(self, n: int, pep_len=(5, 50), rec_len=(50, 254), seed: int = 42):
        self.pep = random_sequences(n, pep_len[0], pep_len[1], seed)
        self.rec = random_sequences(n, rec_len[0], rec_len[1], seed + 1)

    def __len__(self):
        return len(self.pep)

    def __getitem__(self, i):
        return self.pep[i], self.rec[i]
