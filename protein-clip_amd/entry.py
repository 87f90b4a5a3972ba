"""Shared plumbing of the two entry scripts (``main.py``, ``main_2protein_filip.py``): what the
reference's scripts do around the loops — seed, run directory, trunk + tokenizer, loaders,
best-validation checkpoint, ``losses_per_epoch.txt`` (main.py:12-93,
main_2protein_filip.py:11-90) — minus the plotting calls and the dataset download, which are
outside the hot path.

Trunk and data: the reference loads ``facebook/esm2_t30_150M_UR50D`` and downloads its
datasets.  Here the pretrained trunk is used when it is already in the local HuggingFace
cache; otherwise (or with ``--synthetic``) an ESM-2 of the same architecture with random
weights and random residue strings stand in, so the scripts run offline end to end.

Under ``torch.distributed.run`` (WORLD_SIZE > 1) every rank trains on its own pairs and the
contrastive batch spans all ranks (all-gather inside the loss operators).
"""
from __future__ import annotations

import argparse
import os
from datetime import datetime

import torch
import torch.distributed as dist
from torch.utils.data import DataLoader

from . import synthetic
from .modules import seed as seed_mod

ESM_NAME = "facebook/esm2_t30_150M_UR50D"


def parse_args(defaults):
    ap = argparse.ArgumentParser()
    ap.add_argument("--synthetic", action="store_true", help="random-weight trunk and random sequence pairs (no network)")
    ap.add_argument("--epochs", type=int, default=defaults["epochs"])
    ap.add_argument("--batch-size", type=int, default=defaults["batch_size"])
    ap.add_argument("--accumulated-batches", type=int, default=defaults.get("accumulated_batches", 1))
    ap.add_argument("--train-pairs", type=int, default=512, help="synthetic mode: number of training pairs")
    ap.add_argument("--eval-pairs", type=int, default=64, help="synthetic mode: validation / test pairs each")
    ap.add_argument("--trunk-layers", type=int, default=30, help="synthetic mode: depth of the random trunk")
    ap.add_argument("--max-len", type=int, default=254, help="synthetic mode: longest receptor")
    return ap.parse_args()


def init_distributed():
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if torch.cuda.is_available():
        torch.cuda.set_device(local_rank)
    if world > 1 and not dist.is_initialized():
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    return world, int(os.environ.get("RANK", "0")), local_rank


def load_trunk(args, device):
    """(tokenizer, frozen trunk, hidden width, synthetic?)"""
    tokenizer = trunk = None
    if not args.synthetic:
        try:
            from transformers import EsmModel, EsmTokenizer
            tokenizer = EsmTokenizer.from_pretrained(ESM_NAME, local_files_only=True)
            trunk = EsmModel.from_pretrained(ESM_NAME, local_files_only=True)
        except Exception as exc:  # not cached locally and there is no network
            print(f"{ESM_NAME} is not available offline ({type(exc).__name__}); using the random-weight stand-in")
            tokenizer = trunk = None
    is_synth = trunk is None
    if is_synth:
        seed_mod.set_seed()   # every rank of a job builds the same random trunk
        tokenizer = synthetic.make_tokenizer()
        trunk = synthetic.make_esm(640, args.trunk_layers, 20)
    trunk = trunk.to(device)
    for prm in trunk.parameters():
        prm.requires_grad = False
    return tokenizer, trunk, 640, is_synth


def make_loaders(args, rank, generate_datasets=None, pep_len=(5, 50)):
    """Train / validation / test loaders with the reference's settings (shuffle only for
    training, drop_last everywhere; main.py:36-38)."""
    if generate_datasets is not None:
        train_ds, val_ds, test_ds = generate_datasets()
    else:
        rec_len = (50, args.max_len)
        train_ds = synthetic.PairDataset(args.train_pairs, pep_len, rec_len, seed=42 + 1000 * rank)
        val_ds = synthetic.PairDataset(args.eval_pairs, pep_len, rec_len, seed=4242 + 1000 * rank)
        test_ds = synthetic.PairDataset(args.eval_pairs, pep_len, rec_len, seed=424242 + 1000 * rank)
    bs = args.batch_size
    return (DataLoader(train_ds, batch_size=bs, shuffle=True, drop_last=True),
            DataLoader(val_ds, batch_size=bs, shuffle=False, drop_last=True),
            DataLoader(test_ds, batch_size=bs, shuffle=False, drop_last=True))


def run(args, build_model, train_epoch, evaluate, loaders, device, rank):
    """The epoch loop of the reference's scripts: CSV of losses, best-validation checkpoint,
    reload of the best state for the test loss.  Returns (train_losses, val_losses, test_loss)."""
    seed_mod.set_seed()
    distributed = dist.is_available() and dist.is_initialized()
    timestamp = datetime.now().strftime("%Y%m%d_%H%M%S_%f")
    base_path = f"{os.getcwd()}/runs/{timestamp}"
    if distributed:   # one run directory for the job: rank 0's
        shared = [base_path]
        dist.broadcast_object_list(shared, src=0)
        base_path = shared[0]
    if rank == 0:
        os.makedirs(base_path, exist_ok=True)
        print(f"All run info will be saved to {base_path}")
    model = build_model().to(device)
    optimizer = torch.optim.Adam(model.parameters(), lr=1e-3)
    train_loader, val_loader, test_loader = loaders
    model_save_path = f"{base_path}/best_model.pth"
    losses_save_path = f"{base_path}/losses_per_epoch.txt"
    train_losses, val_losses, best_val = [], [], float("inf")
    log = open(losses_save_path, "w") if rank == 0 else None
    if log:
        print(f"Best model will be saved to {model_save_path}")
        print(f"Losses will be saved to {losses_save_path}")
        log.write("Epoch,Train Loss,Validation Loss\n")
    for epoch in range(args.epochs):
        print("new epoch")
        train_loss = train_epoch(model, train_loader, optimizer)
        val_loss = evaluate(model, val_loader)
        train_losses.append(train_loss)
        val_losses.append(val_loss)
        if log:
            log.write(f"{epoch + 1},{train_loss:.4f},{val_loss:.4f}\n")
            log.flush()
            if val_loss < best_val:
                best_val = val_loss
                torch.save(model.state_dict(), model_save_path)
        if torch.cuda.is_available():
            torch.cuda.empty_cache()
        print(f"Epoch {epoch + 1}/{args.epochs} - Train Loss: {train_loss:.4f}, Val Loss: {val_loss:.4f}")
    if log:
        log.close()
    # Test loss of the best state.  The loss operators issue collectives on the whole job, so every
    # rank takes part: all of them load rank 0's checkpoint (written before the barrier) and
    # evaluate their own share of the test pairs; the value is the same on every rank.
    if distributed:
        dist.barrier()
    test_loss = None
    # whether a checkpoint exists is rank 0's to say (it wrote it; the other ranks may not even see its directory):
    # either every rank enters the evaluation's collectives or none does
    have_best = os.path.exists(model_save_path)
    if distributed:
        flag = [have_best]
        dist.broadcast_object_list(flag, src=0)
        have_best = bool(flag[0])
    if have_best:
        best = build_model().to(device)
        if distributed:   # the best state travels from rank 0 (no shared file system is assumed)
            state = [torch.load(model_save_path, weights_only=True, map_location="cpu") if rank == 0 else None]
            dist.broadcast_object_list(state, src=0)
            best.load_state_dict(state[0])
        else:
            best.load_state_dict(torch.load(model_save_path, weights_only=True, map_location=device))
        test_loss = evaluate(best, test_loader)
        if rank == 0:
            print(f"Test Loss: {test_loss:.4f}")
    if distributed:
        dist.barrier()
    return train_losses, val_losses, test_loss
