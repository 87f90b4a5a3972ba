"""Times the pre-allocated CLIP step (pclip_clip_step) for a few batch sizes: events around 200
back-to-back calls, microseconds per step."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from protein_clip_amd import kernels as K  # noqa: E402

for B in [int(a) for a in sys.argv[1:]] or [64, 256, 512, 1024]:
    g = torch.Generator().manual_seed(0)
    e1, e2 = torch.randn(B, 128, generator=g).cuda(), torch.randn(B, 128, generator=g).cuda()
    t = torch.tensor([1.0], device="cuda")
    step = K.ClipStep(B, 128, "cuda")
    for _ in range(50):
        step.run(e1, e2, t)
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(200):
        step.run(e1, e2, t)
    b.record()
    torch.cuda.synchronize()
    print(f"B={B}: {a.elapsed_time(b) / 200 * 1e3:.1f} us per step, loss {step.loss.item():.6f}", flush=True)
    if step.ws.numel() >= 256 + 256 * 128 and os.environ.get("STAMPS"):   # diagnostic build: 16 stamps per workgroup
        nwg = ((B + 31) // 32) ** 2
        st = step.ws[256:256 + nwg * 128].view(torch.int64).cpu().view(nwg, 16).double() / 100.0
        t0 = st[:, 0].min()
        names = ["start", "rows staged", "tile", "statistics", "barrier 1 out", "lse", "slabs stored", "barrier 2 out", "phase C", "loss",
                 "b1 drained", "b1 added", "b1 order", "b2 drained", "b2 added", "b2 order"]
        for k, n in enumerate(names):
            if k in (12, 15):
                continue
            col = st[:, k] - t0
            print(f"   {n:14s} min {col.min():6.2f}  median {col.median():6.2f}  max {col.max():6.2f}   wg0 {col[0]:6.2f}")
