#!/usr/bin/env python3
"""Peptide-receptor CLIP training, the counterpart of the reference's ``main.py``: frozen
ESM-2 trunk, ``ExtendedCLIP(640, 128, 2, 2, 0.1)``, sub-batches of 16 accumulated 16 times into
contrastive batches of 256 (main.py:24-49), Adam 1e-3, 25 epochs, best-validation checkpoint
and ``losses_per_epoch.txt``.  The contrastive step runs on the HIP kernels; with
``torch.distributed.run`` the 256-batch additionally spans every GPU.

    python main.py [--synthetic --epochs 1 --train-pairs 512 --trunk-layers 2]
"""
import os
import sys

import torch
from torch.amp import GradScaler

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import protein_clip_amd  # noqa: E402,F401
from protein_clip_amd import entry  # noqa: E402
from protein_clip_amd.modules import models, training_utils  # noqa: E402


def main():
    args = entry.parse_args({"epochs": 25, "batch_size": 16, "accumulated_batches": 16})
    world, rank, local_rank = entry.init_distributed()
    device = torch.device("cuda", local_rank) if torch.cuda.is_available() else torch.device("cpu")
    tokenizer, esm_model, input_dim, _ = entry.load_trunk(args, device)
    embedding_dim, h1, h2, dropout = 128, 2, 2, 0.1
    loaders = entry.make_loaders(args, rank)
    scaler = GradScaler("cuda", enabled=False)  # the loss kernels are f32: nothing to scale

    def build():
        return models.ExtendedCLIP(input_dim, embedding_dim, h1, h2, dropout, esm_model)

    def train_epoch(model, loader, optimizer):
        return training_utils.train_gc(model, loader, tokenizer, optimizer, scaler, device, args.accumulated_batches)

    def evaluate(model, loader):
        return training_utils.evaluate(model, loader, tokenizer, device)

    return entry.run(args, build, train_epoch, evaluate, loaders, device, rank)


if __name__ == "__main__":
    main()
