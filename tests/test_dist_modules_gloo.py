"""World-size-2 ``gloo`` tests (CPU) of the drop-in surface under ``torch.distributed``:
``modules.training_utils.train``, ``modules.training_utils_filip.train`` / ``_process_batch`` and
``entry.run`` driven exactly as the entry scripts drive them, each rank on its own half of the
pairs.  One optimizer step of the two-rank job must equal the single-process step on the whole
batch: loss, updated head weights (i.e. the summed head gradients) and temperature.  Compute
comes from the test-only CPU provider; the tokenizer is the offline ESM tokenizer and the trunk a
fixed embedding table, so only host logic of the package is under test."""
import argparse
import os
import socket
import sys

import pytest

from conftest import is_port_clash
import torch
import torch.distributed as dist
import torch.multiprocessing as mp
from torch.utils.data import DataLoader

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

WORLD = 2
HID, EMB = 24, 16


class _Out:
    def __init__(self, h):
        self.last_hidden_state = h


class _TableTrunk(torch.nn.Module):
    """Stands in for the frozen ESM trunk: one fixed vector per token id plus a position term."""

    def __init__(self):
        super().__init__()
        g = torch.Generator().manual_seed(11)
        self.table = torch.nn.Parameter(torch.randn(33, HID, generator=g), requires_grad=False)
        self.pos = torch.nn.Parameter(0.1 * torch.randn(400, HID, generator=g), requires_grad=False)

    def forward(self, input_ids, attention_mask):
        return _Out(self.table[input_ids] + self.pos[: input_ids.shape[1]][None])


def _pairs():
    from protein_clip_amd import synthetic as S
    return S.random_sequences(8, 4, 12, seed=5), S.random_sequences(8, 9, 30, seed=6)


class _Pairs(torch.utils.data.Dataset):
    def __init__(self, pep, rec):
        self.pep, self.rec = pep, rec

    def __len__(self):
        return len(self.pep)

    def __getitem__(self, i):
        return self.pep[i], self.rec[i]


def _model(kind):
    from protein_clip_amd.modules import models
    torch.manual_seed(0)
    cls = models.ExtendedCLIP if kind == "clip" else models.ExtendedFILIP
    return cls(HID, EMB, 2, 2, 0.0, _TableTrunk())


def _one_step(kind, pep, rec, tok):
    """One epoch of one batch through the package's own train(); returns (loss, state after the step)."""
    from protein_clip_amd.modules import training_utils, training_utils_filip
    mod = training_utils if kind == "clip" else training_utils_filip
    model = _model(kind)
    opt = torch.optim.SGD([p for p in model.parameters() if p.requires_grad], lr=0.5)
    loader = DataLoader(_Pairs(pep, rec), batch_size=len(pep), shuffle=False)
    loss = mod.train(model, loader, opt, tok, "cpu")
    ev = mod.evaluate(model, loader, tok, "cpu")
    state = {k: v.detach().clone() for k, v in model.state_dict().items() if "esm_model" not in k}
    return loss, ev, state


def _patch_provider():
    import cpu_provider as P
    from protein_clip_amd import ops
    ops._hip_kernels = P


def _worker(rank, port, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(WORLD), LOCAL_RANK=str(rank))
    dist.init_process_group("gloo", rank=rank, world_size=WORLD)
    _patch_provider()
    from protein_clip_amd import entry, synthetic as S
    from protein_clip_amd.modules import models, training_utils_filip
    tok = S.make_tokenizer()
    pep, rec = _pairs()
    half = len(pep) // WORLD
    sl = slice(rank * half, (rank + 1) * half)
    res = {kind: _one_step(kind, pep[sl], rec[sl], tok) for kind in ("clip", "filip")}
    # _process_batch alone: the loss of the rank's batch is the loss of the global batch
    res["filip_batch_loss"] = training_utils_filip._process_batch(_model("filip"), (pep[sl], rec[sl]), tok, "cpu")

    # the epoch loop of the entry scripts, synthetic pairs, one epoch: every rank must come out of it
    # (no rank-0-only collective) with the same test loss
    os.chdir(out_dir)
    args = argparse.Namespace(epochs=1, batch_size=4, train_pairs=8, eval_pairs=4, max_len=60)
    loaders = entry.make_loaders(args, rank, pep_len=(4, 12))
    trunk = _TableTrunk()

    def build():
        return models.ExtendedFILIP(HID, EMB, 2, 2, 0.0, trunk)

    out = entry.run(args, build, lambda m, l, o: training_utils_filip.train(m, l, o, tok, "cpu"),
                    lambda m, l: training_utils_filip.evaluate(m, l, tok, "cpu"), loaders, torch.device("cpu"), rank)
    res["entry"] = out
    torch.save(res, os.path.join(out_dir, f"rank{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


@pytest.fixture(scope="module")
def two_rank_results(tmp_path_factory):
    for attempt in range(3):   # the port is free when it is picked; another process may take it before the ranks meet
        out_dir = str(tmp_path_factory.mktemp("gloo_modules"))   # a fresh directory: a failed attempt may have left files
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        try:
            mp.start_processes(_worker, args=(port, out_dir), nprocs=WORLD, join=True, start_method="spawn")
            break
        except Exception as e:
            if attempt == 2 or not is_port_clash(e):   # only a lost race for the port is worth another attempt
                raise
            print(f"two-rank attempt {attempt}: port {port} was taken, trying another ({e!r})", file=sys.stderr)
    res = [torch.load(os.path.join(out_dir, f"rank{r}.pt"), weights_only=False) for r in range(WORLD)]
    res.append(out_dir)
    return res


@pytest.fixture(scope="module")
def single_results():
    from protein_clip_amd import ops, synthetic as S
    from protein_clip_amd.modules import training_utils_filip
    real = ops._hip_kernels
    _patch_provider()
    try:   # the stand-in provider must not leak into the other tests of this process
        tok = S.make_tokenizer()
        pep, rec = _pairs()
        out = {kind: _one_step(kind, pep, rec, tok) for kind in ("clip", "filip")}
        out["filip_batch_loss"] = training_utils_filip._process_batch(_model("filip"), (pep, rec), tok, "cpu")
    finally:
        ops._hip_kernels = real
    return out


@pytest.mark.parametrize("kind", ["clip", "filip"])
def test_two_rank_train_step_equals_the_single_process_step(two_rank_results, single_results, kind):
    want_loss, want_eval, want_state = single_results[kind]
    for r in range(WORLD):
        loss, ev, state = two_rank_results[r][kind]
        assert abs(loss - want_loss) < 2e-5, (kind, r, loss, want_loss)       # the global-batch loss on every rank
        assert abs(ev - want_eval) < 2e-5, (kind, r, ev, want_eval)           # ... also after the step
        assert set(state) == set(want_state)
        for k in want_state:   # same step on every rank: head gradients and dt were summed over the ranks
            torch.testing.assert_close(state[k], want_state[k], rtol=2e-4, atol=2e-6, msg=lambda m: f"{kind} rank {r} {k}: {m}")
    moved = [k for k in want_state if not torch.equal(want_state[k], _model(kind).state_dict()[k])]
    assert "temperature" in moved and any("projection" in k for k in moved)


def test_process_batch_returns_the_global_loss_on_every_rank(two_rank_results, single_results):
    for r in range(WORLD):
        assert abs(two_rank_results[r]["filip_batch_loss"] - single_results["filip_batch_loss"]) < 2e-5


def test_entry_run_completes_on_every_rank_with_one_test_loss(two_rank_results):
    outs = [two_rank_results[r]["entry"] for r in range(WORLD)]
    for train_losses, val_losses, test_loss in outs:
        assert len(train_losses) == 1 and test_loss is not None and test_loss == test_loss
    assert abs(outs[0][2] - outs[1][2]) < 1e-6 and abs(outs[0][1][0] - outs[1][1][0]) < 1e-6
    runs = os.listdir(os.path.join(two_rank_results[-1], "runs"))
    assert len(runs) == 1   # one run directory for the job
