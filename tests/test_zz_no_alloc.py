"""The allocation-free CLIP step (pclip_clip_step_noalloc) against pclip_clip_step, through the C
ABI: same loss and embedding gradients bit for bit and the same temperature gradient to f32
summation order (rtol 1e-5), on calls in a row of which the second is smaller than the first (the
built-in scratch keeps the previous call's contents), and a clear refusal beyond the built-in size."""
import ctypes

import pytest
import torch

pytestmark = pytest.mark.gpu


def _run_both(lib, B, d, seed):
    from protein_clip_amd._lib import ptr, stream_of
    g = torch.Generator().manual_seed(seed)
    e_pep = torch.randn(B, d, generator=g).cuda()
    e_rec = torch.randn(B, d, generator=g).cuda()
    t = torch.tensor([0.9], device="cuda")
    outs = []
    for noalloc in (False, True):
        loss = torch.full((1,), float("nan"), device="cuda")
        dt = torch.full((1,), float("nan"), device="cuda")
        dp = torch.full((B, d), float("nan"), device="cuda")
        dr = torch.full((B, d), float("nan"), device="cuda")
        if noalloc:
            rc = lib.pclip_clip_step_noalloc(ptr(e_pep), ptr(e_rec), ptr(t), B, d, ptr(loss), ptr(dp), ptr(dr), ptr(dt), stream_of(e_pep))
        else:
            ws = torch.empty(lib.pclip_clip_step_workspace_bytes(B, d), dtype=torch.uint8, device="cuda")
            rc = lib.pclip_clip_step(ptr(e_pep), ptr(e_rec), ptr(t), B, d, ptr(loss), ptr(dp), ptr(dr), ptr(dt), ptr(ws), ws.numel(),
                                     stream_of(e_pep))
        assert rc == 0, lib.pclip_last_error()
        torch.cuda.synchronize()
        outs.append((loss.cpu(), dp.cpu(), dr.cpu(), dt.cpu()))
    return outs, (e_pep.cpu(), e_rec.cpu())


def test_noalloc_clip_step_equals_the_workspace_entry_point():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from oracle import pclip_oracle as O
    from protein_clip_amd import _lib
    lib = _lib.load()
    assert lib.pclip_clip_step_noalloc_max_bytes() == 10 << 20
    for B, d, seed in [(256, 128, 1), (40, 64, 2), (1024, 128, 3), (7, 8, 4)]:   # large, then smaller, then the largest, then tiny
        (want, got), (e_pep, e_rec) = _run_both(lib, B, d, seed)
        for a, b in zip(got[:3], want[:3]):   # loss and both embedding gradients: the same kernels in the same order
            assert torch.isfinite(a).all()
            assert torch.equal(a, b), (B, d, (a - b).abs().max())
        # dt is a sum of float atomic adds, whose order differs from launch to launch
        torch.testing.assert_close(got[3], want[3], rtol=1e-5, atol=1e-7)
        ol, odp, _, odt = O.clip_step(e_pep, e_rec, torch.tensor(0.9))
        torch.testing.assert_close(got[0].reshape(()), ol, rtol=1e-5, atol=1e-6)
        torch.testing.assert_close(got[1], odp, rtol=1e-4, atol=2e-7)
        torch.testing.assert_close(got[3].reshape(()), odt, rtol=1e-4, atol=2e-6)
    one = ctypes.c_void_p(256)
    assert lib.pclip_clip_step_noalloc(one, one, one, 2048, 128, one, one, one, one, None) == -1
    assert b"built-in buffer" in lib.pclip_last_error()
