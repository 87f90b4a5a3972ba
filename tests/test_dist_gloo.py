"""World-size-2 ``gloo`` tests (CPU) of the sharded steps and autograd operators: the
communication pattern (all-gather layout, row offsets, reduce-scatter, scalar all-reduce)
must reproduce the single-process result on the concatenated batch.  Compute comes from the
test-only CPU provider, so only host logic of the package is under test here."""
import os
import sys

import pytest

from conftest import is_port_clash
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

WORLD = int(os.environ.get("PCLIP_TEST_WORLD", "2"))   # 3 puts a rank in the middle (partner blocks before AND behind its own)


def _inputs():
    g = torch.Generator().manual_seed(123)
    B, d, TA, TB = 12, 16, 9, 7
    e_pep, e_rec = torch.randn(B, d, generator=g), torch.randn(B, d, generator=g)
    hA, hB = torch.randn(B, TA, d, generator=g), torch.randn(B, TB, d, generator=g)
    mA = (torch.arange(TA)[None] < torch.randint(2, TA + 1, (B, 1), generator=g)).to(torch.uint8)
    mB = (torch.arange(TB)[None] < torch.randint(2, TB + 1, (B, 1), generator=g)).to(torch.uint8)
    return e_pep, e_rec, hA, hB, mA, mB, torch.tensor([0.9])


def _worker(rank, port, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(WORLD))
    dist.init_process_group("gloo", rank=rank, world_size=WORLD)
    import cpu_provider as P
    from protein_clip_amd import ops
    from protein_clip_amd.steps import ClipShardedStep, FilipShardedStep
    e_pep, e_rec, hA, hB, mA, mB, t = _inputs()
    Bl = e_pep.shape[0] // WORLD
    sl = slice(rank * Bl, (rank + 1) * Bl)
    res = {}
    res["clip_step"] = ClipShardedStep(kernels=P).run(e_pep[sl].contiguous(), e_rec[sl].contiguous(), t)
    res["filip_step"] = FilipShardedStep(kernels=P).run(hA[sl].contiguous(), hB[sl].contiguous(), mA[sl], mB[sl], t)
    # autograd operators with the process group
    x = e_pep[sl].clone().requires_grad_(True)
    y = e_rec[sl].clone().requires_grad_(True)
    tt = torch.tensor(0.9, requires_grad=True)
    loss = ops.clip_loss(ops.l2norm_scale(x, tt, kernels=P), ops.l2norm_scale(y, tt, kernels=P), group=dist.group.WORLD, kernels=P)
    loss.backward()
    res["clip_op"] = (loss.detach(), x.grad, y.grad, tt.grad)
    a = hA[sl].clone().requires_grad_(True)
    b = hB[sl].clone().requires_grad_(True)
    tf = torch.tensor(0.9, requires_grad=True)
    sA, sB = ops.filip_similarity(a, b, mA[sl], mB[sl], tf, normalise=True, group=dist.group.WORLD, kernels=P)
    off, Bg = rank * Bl, WORLD * Bl
    lf = (ops.row_ce(sA, row_offset=off, global_batch=Bg, group=dist.group.WORLD, kernels=P)
          + ops.row_ce(sB, row_offset=off, global_batch=Bg, group=dist.group.WORLD, kernels=P)) / 2
    lf.backward()
    res["filip_op"] = (lf.detach(), a.grad, b.grad, tf.grad)
    torch.save(res, os.path.join(out_dir, f"rank{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


@pytest.fixture(scope="module")
def two_rank_results(tmp_path_factory):
    import socket
    import sys
    for attempt in range(3):   # the port is free when it is picked; another process may take it before the ranks meet
        out_dir = str(tmp_path_factory.mktemp("gloo"))   # a fresh directory: a failed attempt may have left files
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
    return [torch.load(os.path.join(out_dir, f"rank{r}.pt"), weights_only=True) for r in range(WORLD)]


@pytest.fixture(scope="module")
def single_results():
    import cpu_provider as P
    from protein_clip_amd.steps import ClipShardedStep, FilipShardedStep
    e_pep, e_rec, hA, hB, mA, mB, t = _inputs()
    return {
        "clip": ClipShardedStep(kernels=P).run(e_pep, e_rec, t),
        "filip": FilipShardedStep(kernels=P).run(hA, hB, mA, mB, t),
    }


def _cat(results, key, idx):
    return torch.cat([r[key][idx] for r in results])


def test_single_process_cpu_provider_matches_oracle(single_results):
    """The stand-in itself is checked against the oracle, so the sharded comparison below is
    anchored to the reference arithmetic."""
    from oracle import pclip_oracle as O
    e_pep, e_rec, hA, hB, mA, mB, t = _inputs()
    loss, dp, dr, dt = O.clip_step(e_pep, e_rec, t.reshape(()))
    got = single_results["clip"]
    torch.testing.assert_close(got[0].reshape(()), loss, rtol=1e-5, atol=1e-6)
    torch.testing.assert_close(got[1], dp, rtol=1e-4, atol=1e-7)
    torch.testing.assert_close(got[3].reshape(()), dt, rtol=1e-4, atol=1e-6)
    fl = O.filip_step(hA, hB, mA, mB, t.reshape(()))
    torch.testing.assert_close(single_results["filip"][0].reshape(()), fl[0], rtol=5e-3, atol=5e-3)


def test_clip_sharded_step_equals_single_process(two_rank_results, single_results):
    ref = single_results["clip"]
    for r in two_rank_results:
        torch.testing.assert_close(r["clip_step"][0], ref[0], rtol=1e-5, atol=1e-6)   # global loss on every rank
        torch.testing.assert_close(r["clip_step"][3], ref[3], rtol=1e-4, atol=1e-6)   # global dt on every rank
    torch.testing.assert_close(_cat(two_rank_results, "clip_step", 1), ref[1], rtol=1e-4, atol=1e-7)
    torch.testing.assert_close(_cat(two_rank_results, "clip_step", 2), ref[2], rtol=1e-4, atol=1e-7)


def test_filip_sharded_step_equals_single_process(two_rank_results, single_results):
    ref = single_results["filip"]
    for r in two_rank_results:
        torch.testing.assert_close(r["filip_step"][0], ref[0], rtol=1e-5, atol=1e-6)
        torch.testing.assert_close(r["filip_step"][3], ref[3], rtol=1e-4, atol=1e-6)
    torch.testing.assert_close(_cat(two_rank_results, "filip_step", 1), ref[1], rtol=1e-4, atol=1e-7)
    torch.testing.assert_close(_cat(two_rank_results, "filip_step", 2), ref[2], rtol=1e-4, atol=1e-7)


def test_autograd_operators_with_group_equal_steps(two_rank_results, single_results):
    for key, ref in (("clip_op", single_results["clip"]), ("filip_op", single_results["filip"])):
        for r in two_rank_results:
            torch.testing.assert_close(r[key][0].reshape(1), ref[0], rtol=1e-5, atol=1e-6)
        torch.testing.assert_close(_cat(two_rank_results, key, 1), ref[1], rtol=1e-4, atol=1e-7)
        torch.testing.assert_close(_cat(two_rank_results, key, 2), ref[2], rtol=1e-4, atol=1e-7)
        # the temperature gradient of each rank's backward is its share; shares add up to the global one
        total = sum(r[key][3] for r in two_rank_results)
        torch.testing.assert_close(total.reshape(1), ref[3], rtol=1e-4, atol=1e-6, msg=f"{key}: shares {[float(r[key][3]) for r in two_rank_results]} vs {float(ref[3])}")
