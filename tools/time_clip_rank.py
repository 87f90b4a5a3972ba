"""Times pclip_clip_fwd / pclip_clip_bwd at the per-rank CLIP shape of the 8-GPU configuration (1024 own rows
against a gathered batch of 8192), events around 50 back-to-back calls each."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from protein_clip_amd import kernels as K  # noqa: E402

Br, d = 1024, 128
for Bg in [int(v) for v in sys.argv[1:]] or [8192]:
    g = torch.Generator().manual_seed(0)
    x_all = torch.nn.functional.normalize(torch.randn(Bg, d, generator=g), dim=-1).mul(1.6).cuda()
    y_all = torch.nn.functional.normalize(torch.randn(Bg, d, generator=g), dim=-1).mul(1.6).cuda()
    x, y = x_all[:Br].contiguous(), y_all[:Br].contiguous()
    lr, lc, loss, _ = K.clip_fwd(x, y, x_all, y_all, 0)
    lr_all, lc_all = torch.randn(Bg, device="cuda") + 5, torch.randn(Bg, device="cuda") + 5

    def timed(fn, n=50):
        for _ in range(10):
            fn()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(n):
            fn()
        b.record()
        torch.cuda.synchronize()
        return a.elapsed_time(b) / n * 1e3
    f = timed(lambda: K.clip_fwd(x, y, x_all, y_all, 0))
    b = timed(lambda: K.clip_bwd(x, y, x_all, y_all, lr_all, lc_all, 0))
    flops = 2.0 * 2 * Br * Bg * d
    print(f"CLIP {Br}x{Bg}: fwd {f:.1f} us ({flops / f / 1e6:.1f} TFLOP/s), bwd {b:.1f} us ({2 * flops / b / 1e6:.1f} TFLOP/s), loss {loss.item():.5f}", flush=True)
