"""The one-launch CLIP step's grid barrier (csrc/clip_step.hip) through the C ABI: no memset in front of a launch
(the barrier words clear themselves: many launches in a row stay correct), launches on two streams are chained by
the library and complete, a barrier that cannot complete fails VISIBLY (NaN loss and gradients, a sticky flag that
pclip_clip_step_sync_status reports) and pclip_clip_step_sync_reset restores the step, and the allocation-free
entry point replays from a captured graph."""
import ctypes

import pytest
import torch

pytestmark = pytest.mark.gpu


def _buffers(B, d, seed):
    g = torch.Generator().manual_seed(seed)
    e_pep = torch.randn(B, d, generator=g).cuda()
    e_rec = torch.randn(B, d, generator=g).cuda()
    t = torch.tensor([1.1], device="cuda")
    out = dict(loss=torch.full((1,), float("nan"), device="cuda"), dt=torch.full((1,), float("nan"), device="cuda"),
               dp=torch.full((B, d), float("nan"), device="cuda"), dr=torch.full((B, d), float("nan"), device="cuda"))
    return e_pep, e_rec, t, out


def _step(lib, e_pep, e_rec, t, out, ws, stream):
    from protein_clip_amd._lib import ptr
    B, d = e_pep.shape
    rc = lib.pclip_clip_step(ptr(e_pep), ptr(e_rec), ptr(t), B, d, ptr(out["loss"]), ptr(out["dp"]), ptr(out["dr"]), ptr(out["dt"]),
                             ptr(ws), ws.numel(), ctypes.c_void_p(stream.cuda_stream))
    assert rc == 0, lib.pclip_last_error()


def _check(out, e_pep, e_rec, t):
    from oracle import pclip_oracle as O
    ol, odp, odr, odt = O.clip_step(e_pep.cpu(), e_rec.cpu(), t.cpu().reshape(()))
    torch.testing.assert_close(out["loss"].cpu().reshape(()), ol, rtol=1e-5, atol=1e-6)
    torch.testing.assert_close(out["dp"].cpu(), odp, rtol=1e-4, atol=2e-7)
    torch.testing.assert_close(out["dr"].cpu(), odr, rtol=1e-4, atol=2e-7)
    torch.testing.assert_close(out["dt"].cpu().reshape(()), odt, rtol=1e-4, atol=2e-6)


@pytest.fixture()
def lib():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from protein_clip_amd import _lib
    lib = _lib.load()
    gave_up = ctypes.c_int(-1)
    assert lib.pclip_clip_step_sync_status(ctypes.byref(gave_up)) == 0 and gave_up.value == 0
    yield lib
    lib.pclip_clip_step_selftest_giveup(0)
    lib.pclip_clip_step_sync_reset(None)
    torch.cuda.synchronize()


def test_two_streams_complete_and_match(lib):
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    torch.cuda.synchronize()
    jobs = []
    for k, (B, d) in enumerate([(256, 128), (512, 128), (96, 64), (256, 128)]):
        e_pep, e_rec, t, out = _buffers(B, d, 10 + k)
        ws = torch.empty(lib.pclip_clip_step_workspace_bytes(B, d), dtype=torch.uint8, device="cuda")
        jobs.append((e_pep, e_rec, t, out, ws))
    torch.cuda.synchronize()
    for rep in range(8):   # alternating streams, nothing synchronised in between: the library chains the launches
        for k, job in enumerate(jobs):
            _step(lib, *job, s1 if (k + rep) % 2 == 0 else s2)
    s1.synchronize()
    s2.synchronize()
    for e_pep, e_rec, t, out, _ in jobs:
        _check(out, e_pep, e_rec, t)
    gave_up = ctypes.c_int(-1)
    assert lib.pclip_clip_step_sync_status(ctypes.byref(gave_up)) == 0 and gave_up.value == 0


def test_barrier_that_cannot_complete_fails_visibly(lib):
    e_pep, e_rec, t, out = _buffers(64, 128, 3)
    ws = torch.empty(lib.pclip_clip_step_workspace_bytes(64, 128), dtype=torch.uint8, device="cuda")
    st = torch.cuda.current_stream()
    _step(lib, e_pep, e_rec, t, out, ws, st)
    torch.cuda.synchronize()
    _check(out, e_pep, e_rec, t)
    lib.pclip_clip_step_selftest_giveup(1)   # every wait now lacks one arrival
    _step(lib, e_pep, e_rec, t, out, ws, st)
    torch.cuda.synchronize()
    assert torch.isnan(out["loss"]).all() and torch.isnan(out["dp"]).all() and torch.isnan(out["dr"]).all() and torch.isnan(out["dt"]).all()
    gave_up = ctypes.c_int(0)
    assert lib.pclip_clip_step_sync_status(ctypes.byref(gave_up)) == 0 and gave_up.value == 1
    lib.pclip_clip_step_selftest_giveup(0)   # the flag is sticky: a healthy launch still reports the failure
    for v in out.values():
        v.zero_()
    _step(lib, e_pep, e_rec, t, out, ws, st)
    torch.cuda.synchronize()
    assert torch.isnan(out["loss"]).all() and torch.isnan(out["dp"]).all()
    assert lib.pclip_clip_step_sync_reset(None) == 0
    torch.cuda.synchronize()
    _step(lib, e_pep, e_rec, t, out, ws, st)
    torch.cuda.synchronize()
    _check(out, e_pep, e_rec, t)


def test_noalloc_step_replays_from_a_captured_graph(lib):
    from protein_clip_amd._lib import ptr
    B, d = 256, 128
    e_pep, e_rec, t, out = _buffers(B, d, 5)

    def call(stream):
        rc = lib.pclip_clip_step_noalloc(ptr(e_pep), ptr(e_rec), ptr(t), B, d, ptr(out["loss"]), ptr(out["dp"]), ptr(out["dr"]), ptr(out["dt"]),
                                         ctypes.c_void_p(stream.cuda_stream))
        assert rc == 0, lib.pclip_last_error()

    call(torch.cuda.current_stream())   # the eager call that looks the scratch address and the occupancy up
    torch.cuda.synchronize()
    _check(out, e_pep, e_rec, t)
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        call(torch.cuda.current_stream())
    for rep in range(3):
        e_pep.copy_(torch.randn(B, d, generator=torch.Generator().manual_seed(50 + rep)))
        for v in out.values():
            v.fill_(float("nan"))
        graph.replay()
        torch.cuda.synchronize()
        _check(out, e_pep, e_rec, t)
