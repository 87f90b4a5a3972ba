"""The sharded steps and operators through the REAL collective backend (RCCL) in a process
group of one rank: every all-gather / reduce-scatter / all-reduce the multi-GPU path issues is
executed with its actual dtypes and shapes (the bf16 token images travel as int16 bit patterns,
which RCCL has no type for), and the results must equal the path without collectives."""
import socket

import pytest
import torch
import torch.distributed as dist

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def one_rank_group():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from protein_clip_amd import ops, steps
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    ops.FORCE_COLLECTIVES = steps.FORCE_COLLECTIVES = True
    yield
    ops.FORCE_COLLECTIVES = steps.FORCE_COLLECTIVES = False
    dist.destroy_process_group()


def _inputs():
    g = torch.Generator().manual_seed(4)
    B, T, d = 12, 70, 128
    hA, hB = torch.randn(B, T, d, generator=g).cuda(), torch.randn(B, T, d, generator=g).cuda()
    mA = (torch.arange(T)[None] < torch.randint(20, T + 1, (B, 1), generator=g)).to(torch.uint8).cuda()
    mB = (torch.arange(T)[None] < torch.randint(20, T + 1, (B, 1), generator=g)).to(torch.uint8).cuda()
    e1, e2 = torch.randn(40, d, generator=g).cuda(), torch.randn(40, d, generator=g).cuda()
    return hA, hB, mA, mB, e1, e2, torch.tensor([1.3], device="cuda")


def test_sharded_steps_through_rccl_equal_the_local_path(one_rank_group):
    from protein_clip_amd import steps
    hA, hB, mA, mB, e1, e2, t = _inputs()
    got_f = steps.FilipShardedStep().run(hA, hB, mA, mB, t)
    got_c = steps.ClipShardedStep().run(e1, e2, t)
    steps.FORCE_COLLECTIVES = False
    try:
        ref_f = steps.FilipShardedStep().run(hA, hB, mA, mB, t)
        ref_c = steps.ClipShardedStep().run(e1, e2, t)
    finally:
        steps.FORCE_COLLECTIVES = True
    for a, b in zip(got_f + got_c, ref_f + ref_c):
        torch.testing.assert_close(a, b, rtol=1e-5, atol=1e-7)


def test_autograd_operators_through_rccl_equal_the_local_path(one_rank_group):
    from protein_clip_amd import ops
    hA, hB, mA, mB, e1, e2, t = _inputs()

    def run():
        a, b, tt = hA.clone().requires_grad_(True), hB.clone().requires_grad_(True), t.clone().requires_grad_(True)
        sA, sB = ops.filip_similarity(a, b, mA, mB, tt, normalise=True, group=dist.group.WORLD)
        loss = (ops.row_ce(sA) + ops.row_ce(sB)) / 2
        x, y = e1.clone().requires_grad_(True), e2.clone().requires_grad_(True)
        closs = ops.clip_loss(ops.l2norm_scale(x, tt), ops.l2norm_scale(y, tt), group=dist.group.WORLD)
        (loss + closs).backward()
        return loss.detach(), closs.detach(), a.grad, b.grad, x.grad, y.grad, tt.grad

    got = run()
    ops.FORCE_COLLECTIVES = False
    try:
        ref = run()
    finally:
        ops.FORCE_COLLECTIVES = True
    for a, b in zip(got, ref):
        torch.testing.assert_close(a, b, rtol=1e-5, atol=1e-7)


def test_bench_multi_rank_path_rehearsal():
    """bench.py through its N>1 code (process group, barriers, sharded steps with every
    collective, max-over-ranks timing) in a one-rank RCCL group, as a child process."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    res = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--steps", "2", "--warmup", "1", "--rehearse-dist"],
                         capture_output=True, text=True, timeout=300, cwd=root)
    assert res.returncode == 0, res.stderr[-2000:]
    line = [l for l in res.stdout.splitlines() if l.startswith("{")][-1]
    out = json.loads(line)
    assert out["config"]["clip_global_batch"] == 1024 and out["value"] > 0 and "rehearsal" in out["note"]
    assert out["device_allocations_in_timed_steps"] == 0   # the sharded steps run on buffers made once
