#!/usr/bin/env python3
"""Times, on ONE GPU, the kernels one rank of an N-GPU job runs (no collectives): FILIP with
256 own sequences against 256*N gathered ones, CLIP with 1024 own rows against 1024*N.
    python3 tools/run_rank_shapes.py [N ...]
"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from protein_clip_amd import kernels as K  # noqa: E402


def timed(fn, reps=5):
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3


def main():
    dev = torch.device("cuda", 0)
    t = torch.tensor([1.0], device=dev)
    for n in [int(a) for a in sys.argv[1:]] or [1, 2, 4, 8]:
        g = torch.Generator().manual_seed(n)
        Bl, L, d = 256, 256, 128
        Bg = Bl * n
        hA = torch.randn(Bl, L, d, generator=g).to(dev)
        hB = torch.randn(Bg, L, d, generator=g).to(dev)
        mA = (torch.arange(L)[None] < torch.randint(128, L + 1, (Bl, 1), generator=g)).to(torch.uint8).to(dev)
        mB = (torch.arange(L)[None] < torch.randint(128, L + 1, (Bg, 1), generator=g)).to(torch.uint8).to(dev)
        hnA, mwA = K.filip_prep(hA, mA, True, tsign=t)
        hnB, mwB = K.filip_prep(hB, mB, True)
        out = {}

        def fwd():
            out["f"] = K.filip_sim_fwd(hnA, mwA, L, hnB, mwB, L, t, want_arg=True, unit_norm=True)
        ms_f = timed(fwd)
        simA, simB, argA, argB = out["f"]
        gA = torch.randn(Bl, Bg, device=dev) / Bg
        gB = torch.randn(Bl, Bg, device=dev) / Bg
        ms_b = timed(lambda: K.filip_sim_bwd(hnA, mwA, L, hnB, mwB, L, t, simA, simB, argA, argB, gA, gB))
        flops = 2.0 * d * float(mA.sum()) * float(mB.sum())
        Cl = 1024
        Cg = Cl * n
        x = torch.nn.functional.normalize(torch.randn(Cl, d, generator=g)).to(dev) * 1.6487
        y_all = torch.nn.functional.normalize(torch.randn(Cg, d, generator=g)).to(dev) * 1.6487
        x_all = torch.nn.functional.normalize(torch.randn(Cg, d, generator=g)).to(dev) * 1.6487
        y = y_all[:Cl].contiguous()
        x_all[:Cl] = x
        res = {}

        def cf():
            res["r"] = K.clip_fwd(x, y, x_all, y_all, 0)
        ms_cf = timed(cf, 20)
        lse_r, lse_c = res["r"][0], res["r"][1]
        lr = lse_r.repeat(n)
        lc = lse_c.repeat(n)
        ms_cb = timed(lambda: K.clip_bwd(x, y, x_all, y_all, lr, lc, 0), 20)
        print(f"N={n}: FILIP 256x{Bg}: scores {ms_f:.3f} ms ({flops / ms_f / 1e9:.0f} TFLOP/s), backward {ms_b:.3f} ms | "
              f"CLIP 1024x{Cg}: fwd {ms_cf:.3f} ms, bwd {ms_cb:.3f} ms", flush=True)


if __name__ == "__main__":
    main()
