#!/usr/bin/env python3
"""Protein-protein FILIP training, the counterpart of the reference's
``main_2protein_filip.py``: frozen ESM-2 trunk, ``ExtendedFILIP(640, 128, 2, 2, 0.1)``, batch 10,
Adam 1e-3, 10 epochs (main_2protein_filip.py:23-45), best-validation checkpoint and
``losses_per_epoch.txt``.  Token normalisation, the token-wise max-similarity scores and both
row InfoNCE losses run on the HIP kernels (bf16 matrix cores), so the batch is no longer
limited by a materialised B x B x T x T tensor: ``--batch-size 256`` works.

    python main_2protein_filip.py [--synthetic --epochs 1 --batch-size 32 --trunk-layers 2]
"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import protein_clip_amd  # noqa: E402,F401
from protein_clip_amd import entry  # noqa: E402
from protein_clip_amd.modules import models, training_utils_filip  # noqa: E402


def main():
    args = entry.parse_args({"epochs": 10, "batch_size": 10})
    world, rank, local_rank = entry.init_distributed()
    device = torch.device("cuda", local_rank) if torch.cuda.is_available() else torch.device("cpu")
    tokenizer, esm_model, input_dim, _ = entry.load_trunk(args, device)
    embedding_dim, h1, h2, dropout = 128, 2, 2, 0.1
    loaders = entry.make_loaders(args, rank, pep_len=(50, args.max_len))   # both chains are proteins

    def build():
        return models.ExtendedFILIP(input_dim, embedding_dim, h1, h2, dropout, esm_model)

    def train_epoch(model, loader, optimizer):
        return training_utils_filip.train(model, loader, optimizer, tokenizer, device)

    def evaluate(model, loader):
        return training_utils_filip.evaluate(model, loader, tokenizer, device)

    return entry.run(args, build, train_epoch, evaluate, loaders, device, rank)


if __name__ == "__main__":
    main()
