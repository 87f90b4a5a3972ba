#!/usr/bin/env python3
"""Benchmark of the contrastive hot path on MI355X.

    python bench.py [--gpus N --steps K --warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

One "step" = one pass of BOTH workloads the headline metric names, forward + backward from
raw head outputs to their gradients, inputs resident in HBM:
  * CLIP  — per-GPU batch 256 (N=1; global batch 256) or 1024 (N>1; global 1024*N, i.e. 8192
            on 8 GPUs), d = 128, f32: normalise*exp(t/2) -> symmetric InfoNCE -> backward;
  * FILIP — per-GPU batch 256 (global 256*N, i.e. 2048 on 8 GPUs), L = 256 tokens (valid
            lengths U[128,256]), d = 128: token normalise -> bf16-MFMA max-similarity ->
            two row InfoNCE losses -> backward.
``value`` = protein pairs of both global batches per second of whole-job time (max over
ranks).  The per-workload numbers and rooflines are in ``workloads``.  Rank 0 prints ONE JSON
line.  The CPU baseline (N=1 only) times the repository's CPU oracle on the same inputs: the
whole step of both workloads on all host threads (about half a minute), one-thread figures beside it.
"""
import argparse
import json
import os
import statistics
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

D = 128
L = 256
SETTLE_STEPS = 30   # untimed steps in front of the warm-up (clock ramp), reported in the JSON line
PEAK_BF16_TFLOPS = 2500.0   # dense bf16 MFMA peak, MI355X
PEAK_F32_MFMA_TFLOPS = 157.3
PEAK_HBM_GBS = 8000.0


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--workload", choices=["both", "clip", "filip"], default="both")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--full-length", action="store_true",
                    help="also time the score kernel alone with every token valid (secondary record; off by default so that a "
                         "profiler trace of the default command holds launches of the benchmark shape only)")
    ap.add_argument("--no-rank-shapes", action="store_true",
                    help="skip the secondary records that time, on this one GPU, the kernels one rank of the two 8-GPU "
                         "configurations runs (CLIP 1024 x 8192, FILIP 256 x 2048; never part of `value`)")
    ap.add_argument("--rehearse-dist", action="store_true",
                    help="one GPU, but through the multi-rank code: a one-rank RCCL group, the sharded steps with every "
                         "collective issued, the per-rank shapes of an N>1 run (not a benchmark result)")
    return ap.parse_args()


def make_inputs(rank, clip_bl, filip_bl, device):
    g = torch.Generator().manual_seed(rank)          # rank 0 == the single-GPU inputs
    e_pep = torch.randn(clip_bl, D, generator=g)
    e_rec = torch.randn(clip_bl, D, generator=g)
    hA = torch.randn(filip_bl, L, D, generator=g).to(torch.bfloat16).float()
    hB = torch.randn(filip_bl, L, D, generator=g).to(torch.bfloat16).float()
    lA = torch.randint(128, L + 1, (filip_bl,), generator=g)
    lB = torch.randint(128, L + 1, (filip_bl,), generator=g)
    ar = torch.arange(L)[None]
    mA = (ar < lA[:, None]).to(torch.uint8)
    mB = (ar < lB[:, None]).to(torch.uint8)
    cpu = dict(e_pep=e_pep, e_rec=e_rec, hA=hA, hB=hB, mA=mA, mB=mB, lA=lA, lB=lB)
    dev = {k: v.to(device).contiguous() for k, v in cpu.items()}
    return cpu, dev


def cpu_baseline(cpu, clip_b, filip_b):
    """Oracle (PyTorch-CPU restatement of the reference path) on the host cores, same inputs and the same step (forward
    + backward from raw head outputs to their gradients): CLIP at the benchmark batch (median of 10) and the FILIP
    B = 256 step run IN FULL once, in 16 x 16 blocks of sequence pairs (about 25 s on 16 threads); that is `value`.
    Beside it: CLIP at the 8-GPU global batch of 8192, and one-thread figures (CLIP in full; FILIP on the first
    32 x 32 sequence pairs, scaled by (B/32)^2 and marked as extrapolated)."""
    from oracle import pclip_oracle as O
    t = torch.tensor(1.0)
    threads = torch.get_num_threads()

    def clip_ms(B, reps, e_pep=None, e_rec=None):
        if e_pep is None:
            g = torch.Generator().manual_seed(0)
            e_pep, e_rec = torch.randn(B, D, generator=g), torch.randn(B, D, generator=g)
        ts = []
        for _ in range(reps):
            t0 = time.perf_counter()
            O.clip_step(e_pep[:B], e_rec[:B], t)
            ts.append(time.perf_counter() - t0)
        return statistics.median(ts[min(2, reps - 1):]) * 1e3

    def filip_s(sb):
        t0 = time.perf_counter()
        O.filip_step(cpu["hA"][:sb], cpu["hB"][:sb], cpu["mA"][:sb], cpu["mB"][:sb], t, block=16)
        return time.perf_counter() - t0

    t_clip = clip_ms(clip_b, 12, cpu["e_pep"], cpu["e_rec"])
    t_clip_8192 = clip_ms(8192, 3)
    t_filip = filip_s(filip_b) * 1e3
    torch.set_num_threads(1)
    try:
        t_clip_1 = clip_ms(clip_b, 12, cpu["e_pep"], cpu["e_rec"])
        t_clip_8192_1 = clip_ms(8192, 1)
        sb = 32
        t_filip_1 = filip_s(sb) * (filip_b / sb) ** 2 * 1e3
    finally:
        torch.set_num_threads(threads)
    return {
        "value": (clip_b + filip_b) / ((t_clip + t_filip) * 1e-3),
        "unit": "pairs/s",
        "cores": threads,
        "kind": "port",
        "sample": f"CLIP B={clip_b} full step, median of 10 ({t_clip:.3f} ms); FILIP B={filip_b} L={L} full step, run once in "
                  f"16x16 blocks of sequence pairs ({t_filip / 1e3:.2f} s)",
        "extrapolated": False,
        "clip_ms": t_clip,
        "filip_ms": t_filip,
        "clip_b8192_ms": t_clip_8192,
        "one_thread": {
            "clip_ms": t_clip_1, "clip_b8192_ms": t_clip_8192_1,
            "filip_ms": t_filip_1, "filip_extrapolated": True,
            "filip_sample": f"first {sb}x{sb} sequence pairs (1/{(filip_b // sb) ** 2} of the step) scaled by {(filip_b // sb) ** 2}",
            "value": (clip_b + filip_b) / ((t_clip_1 + t_filip_1) * 1e-3),
        },
    }


def _median_ms(fn, reps):
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
    fn()
    for a, b in ev:
        a.record()
        fn()
        b.record()
    torch.cuda.synchronize()
    return statistics.median(a.elapsed_time(b) for a, b in ev)


def rank_shape_records(K, device):
    """One rank's kernels of the two 8-GPU configurations, timed on this GPU with the gathered
    batch formed locally (no collectives): what the all-gather would deliver, then exactly the
    launches a rank issues."""
    out = {}
    g = torch.Generator().manual_seed(88)
    t = torch.tensor([1.0], device=device)
    # CLIP: 1024 own rows of a global batch of 8192
    Br, Bg = 1024, 8192
    x_all = torch.nn.functional.normalize(torch.randn(Bg, D, generator=g)).to(device) * 1.6487
    y_all = torch.nn.functional.normalize(torch.randn(Bg, D, generator=g)).to(device) * 1.6487
    off = 3072
    x, y = x_all[off:off + Br].contiguous(), y_all[off:off + Br].contiguous()
    res = {}

    def cf():
        res["r"] = K.clip_fwd(x, y, x_all, y_all, off)
    ms_f = _median_ms(cf, 20)
    lr, lc = res["r"][0].repeat(Bg // Br), res["r"][1].repeat(Bg // Br)   # stand-ins for the gathered vectors
    ms_b = _median_ms(lambda: K.clip_bwd(x, y, x_all, y_all, lr, lc, off), 20)
    flop = 6.0 * Br * Bg * D
    out["rank_shape_clip_1024x8192"] = {
        "note": "one rank of CLIP on 8 GPUs (global batch 8192), kernels only, timed on one GPU; not part of `value`",
        "fwd_ms": ms_f, "bwd_ms": ms_b, "algorithmic_flops": flop,
        "f32_mfma_TFLOPs": flop / ((ms_f + ms_b) * 1e-3) / 1e12,
        "f32_mfma_frac": flop / ((ms_f + ms_b) * 1e-3) / 1e12 / PEAK_F32_MFMA_TFLOPS,
    }
    # FILIP: 256 own A sequences against 2048 gathered B sequences
    Bl, Bq = 256, 2048
    hA = torch.randn(Bl, L, D, generator=g).to(device)
    hB = torch.randn(Bq, L, D, generator=g).to(device)
    lA = torch.randint(128, L + 1, (Bl, 1), generator=g)
    lB = torch.randint(128, L + 1, (Bq, 1), generator=g)
    mA = (torch.arange(L)[None] < lA).to(torch.uint8).to(device)
    mB = (torch.arange(L)[None] < lB).to(torch.uint8).to(device)
    hnA, mwA = K.filip_prep(hA, mA, True, tsign=t)
    hnB, mwB = K.filip_prep(hB, mB, True)
    keep = {}

    def ff():
        keep["f"] = K.filip_sim_fwd(hnA, mwA, L, hnB, mwB, L, t, want_arg=True, unit_norm=True)
    ms_s = _median_ms(ff, 5)
    simA, simB, argA, argB = keep["f"]
    gA = torch.randn(Bl, Bq, device=device) / Bq
    gB = torch.randn(Bl, Bq, device=device) / Bq
    ms_bw = _median_ms(lambda: K.filip_sim_bwd(hnA, mwA, L, hnB, mwB, L, t, simA, simB, argA, argB, gA, gB), 5)
    fl = 2.0 * D * float(lA.sum()) * float(lB.sum())
    out["rank_shape_filip_256x2048"] = {
        "note": "one rank of FILIP on 8 GPUs (global batch 2048), kernels only, timed on one GPU; not part of `value`",
        "scores_ms": ms_s, "backward_ms": ms_bw, "algorithmic_flops_scores": fl,
        "sim_kernel_TFLOPs": fl / (ms_s * 1e-3) / 1e12, "sim_kernel_mfma_frac": fl / (ms_s * 1e-3) / 1e12 / PEAK_BF16_TFLOPS,
    }
    return out


def library_gemm_tflops(device, n=8192, reps=20):
    """Rate of the library's plain bf16 GEMM (torch.matmul -> hipBLASLt) on random operands, measured on this GPU:
    the matrix cores' sustained rate depends on the operand data (with all-zero operands the same GEMM runs a third
    faster), so the nominal peak is not reachable on real data and this is the practical ceiling beside it."""
    g = torch.Generator(device="cpu").manual_seed(7)
    a = torch.randn(n, n, generator=g).to(device=device, dtype=torch.bfloat16)
    b = torch.randn(n, n, generator=g).to(device=device, dtype=torch.bfloat16)
    for _ in range(3):
        a @ b
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(reps):
        a @ b
    e1.record()
    torch.cuda.synchronize()
    return 2.0 * n ** 3 / (e0.elapsed_time(e1) / reps * 1e-3) / 1e12


def long_chain_record(K, device):
    """The FILIP scores of a batch of long chains (the reference keeps chains of up to 2,000 residues,
    modules/data_utils_2protein.py:88,133; its FILIP entry script runs batches of 10): 16 x 16 sequences of 1,000
    tokens and 10 x 10 of 2,000, every token valid, the call as a whole (256-token segments on the streaming
    kernel + the joining kernel), timed with events after the main benchmark (warm clocks)."""
    out = {}
    t = torch.tensor([1.0], device=device)
    for B, T in ((16, 1000), (10, 2000)):
        g = torch.Generator().manual_seed(T)
        hA = torch.randn(B, T, D, generator=g).to(device)
        hB = torch.randn(B, T, D, generator=g).to(device)
        ones = torch.ones(B, T, dtype=torch.uint8, device=device)
        hnA, mwA = K.filip_prep(hA, ones, True, tsign=t)
        hnB, mwB = K.filip_prep(hB, ones, True)
        for _ in range(20):
            K.filip_sim_fwd(hnA, mwA, T, hnB, mwB, T, t, want_arg=True, unit_norm=True)
        ms = _median_ms(lambda: K.filip_sim_fwd(hnA, mwA, T, hnB, mwB, T, t, want_arg=True, unit_norm=True), 40)
        fl = 2.0 * D * (B * T) ** 2
        out[f"filip_long_b{B}_t{T}"] = {
            "note": "scores of long chains, every token valid: segments of 256 tokens on the streaming kernel, then the join (not part of `value`)",
            "scores_ms": ms, "algorithmic_flops": fl, "TFLOPs": fl / (ms * 1e-3) / 1e12, "mfma_frac": fl / (ms * 1e-3) / 1e12 / PEAK_BF16_TFLOPS}
    return out


def head_record(K, device):
    """The step in front of the hot path: masked mean over tokens and the embedding head of one CLIP tower
    (Linear -> ReLU -> LayerNorm -> Dropout -> Linear -> normalise, scale), forward and backward kernels
    alone at B = 256, T = 256, d = 128."""
    B, T = 256, 256
    g = torch.Generator().manual_seed(5)
    h = torch.randn(B, T, D, generator=g).to(device)
    mask = (torch.arange(T)[None] < torch.randint(128, T + 1, (B, 1), generator=g)).to(torch.uint8).to(device)
    params = [(torch.randn(D, D, generator=g) / D ** 0.5).to(device), torch.zeros(D, device=device), torch.ones(D, device=device),
              torch.zeros(D, device=device), (torch.randn(D, D, generator=g) / D ** 0.5).to(device), torch.zeros(D, device=device)]
    keep = (torch.rand(1, B, D, generator=g) >= 0.1).to(torch.uint8).to(device)
    t = torch.tensor([1.0], device=device)
    w = torch.randn(B, D, generator=g).to(device)
    st = {}

    def fwd():
        st["pool"] = K.masked_mean_fwd(h, mask)
        st["out"] = K.head_tail_fwd(st["pool"][0], params, keep, 1.0 / 0.9, 1e-5, t)

    def bwd():
        pooled, inv, m8 = st["pool"]
        d_pooled, _, _ = K.head_tail_bwd(pooled, params, keep, 1.0 / 0.9, 1e-5, t, st["out"][1], w)
        K.masked_mean_bwd(d_pooled, m8, inv, T)
    ms_f = _median_ms(fwd, 30)
    ms_b = _median_ms(bwd, 30)
    byts = 2.0 * B * T * D * 4   # the token tensor read once forward, its gradient written once backward
    return {"embedding_head_b256_t256": {
        "note": "masked mean + embedding head of one CLIP tower, kernels only (host wrappers included); not part of `value`",
        "fwd_ms": ms_f, "bwd_ms": ms_b, "algorithmic_bytes": byts, "hbm_GBs": byts / ((ms_f + ms_b) * 1e-3) / 1e9,
        "hbm_frac": byts / ((ms_f + ms_b) * 1e-3) / 1e9 / PEAK_HBM_GBS}}


def main():
    args = parse()
    # Libraries below this script write to file descriptor 1 on their own (RCCL prints a version banner there when
    # a process group is created).  The contract is ONE JSON line on stdout, so everything else is sent to stderr
    # and the line is written to the real stdout at the end.
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with torch.distributed.run --nproc-per-node N for --gpus N > 1")
        args.gpus = world
    import torch.distributed as dist
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    rehearse = args.rehearse_dist and world == 1
    if world > 1:
        dist.init_process_group("nccl", device_id=device)
    elif rehearse:
        import socket
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1, device_id=device)
        from protein_clip_amd import steps as _steps
        _steps.FORCE_COLLECTIVES = True

    from protein_clip_amd import _lib, kernels as K
    from protein_clip_amd.steps import ClipShardedStep, FilipShardedStep
    lib = _lib.load()
    if lib.pclip_check_device(local_rank) != 0:
        raise SystemExit(lib.pclip_last_error().decode())

    multi = world > 1 or rehearse          # the multi-rank code paths and shapes
    clip_bl = 1024 if multi else 256
    filip_bl = 256
    cpu, dv = make_inputs(rank, clip_bl, filip_bl, device)
    t = torch.tensor([1.0], device=device)

    # one GPU: the pre-allocated steps (every buffer created once, a step only enqueues kernels);
    # several ranks: the sharded steps with their collectives
    clip_fused = None if multi else K.ClipStep(clip_bl, D, device)
    clip_sh = ClipShardedStep() if multi else None
    filip_fused = None if multi else K.FilipStep(filip_bl, filip_bl, L, L, D, device)
    filip_sh = FilipShardedStep() if multi else None

    def clip_step():
        if clip_fused is not None:
            return clip_fused.run(dv["e_pep"], dv["e_rec"], t)
        return clip_sh.run(dv["e_pep"], dv["e_rec"], t)

    def filip_step(ev=None, bev=None):
        if filip_fused is not None:
            return filip_fused.run(dv["hA"], dv["hB"], dv["mA"], dv["mB"], t, sim_events=ev, bwd_events=bev)
        return filip_sh.run(dv["hA"], dv["hB"], dv["mA"], dv["mB"], t, sim_events=ev)

    allocs = {"n": 0}

    def bracket(fn, steps):
        """Barrier + sync, run, sync + barrier; returns seconds (max over ranks)."""
        torch.cuda.synchronize()
        if multi:
            dist.barrier()
        torch.cuda.synchronize()
        a0 = torch.cuda.memory_stats(device).get("allocation.all.allocated", 0)
        t0 = time.perf_counter()
        for i in range(steps):
            fn(i)
        allocs["n"] = torch.cuda.memory_stats(device).get("allocation.all.allocated", 0) - a0   # by the steps themselves
        torch.cuda.synchronize()
        if multi:
            dist.barrier()
        el = time.perf_counter() - t0
        if multi:
            tt = torch.tensor([el], device=device, dtype=torch.float64)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            el = tt.item()
        return el

    do_clip = args.workload in ("both", "clip")
    do_filip = args.workload in ("both", "filip")
    events = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]
    bwd_events = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]

    def step(i, ev=True):
        if do_clip:
            clip_step()
        if do_filip:
            filip_step(events[i] if ev else None, bwd_events[i] if ev else None)

    # Clock settle: an idle MI355X needs about 15 steps of this size to reach its steady clocks
    # (the score kernel goes from 0.82 to 0.73 ms over them).  A fixed number of untimed steps
    # in front of the caller's W warm-up steps, so that a small W does not time the ramp.
    for i in range(SETTLE_STEPS):
        step(0, ev=False)
    for i in range(args.warmup):
        step(i, ev=False)
    total = bracket(lambda i: step(i), args.steps)
    step_allocs = allocs["n"]   # device allocations made by the timed steps
    if multi and step_allocs != 0:
        msg = f"the sharded steps allocated device memory {step_allocs} times inside the timed region (expected: none)"
        if rehearse:   # the rehearsal is the check; a real multi-GPU run reports the count in its JSON line and goes on
            raise SystemExit(msg)
        print("bench.py: " + msg, file=sys.stderr)
    sim_ms = statistics.mean(e0.elapsed_time(e1) for e0, e1 in events) if do_filip else None
    bwd_ms = statistics.mean(e0.elapsed_time(e1) for e0, e1 in bwd_events) if do_filip and filip_fused is not None else None

    # per-workload brackets (same step counts), for the nested records
    t_clip = bracket(lambda i: clip_step(), args.steps) if do_clip else None
    t_filip = bracket(lambda i: filip_step(None), args.steps) if do_filip else None

    # secondary: the score kernel alone on full-length sequences (all 256 tokens valid), the
    # shape on which its distance from the MFMA peak is cleanest to read (no padding effects)
    full_ms = None
    if do_filip and not multi and args.full_length:
        ones = torch.ones(filip_bl, L, dtype=torch.uint8, device=device)
        hnA_f, mwA_f = K.filip_prep(dv["hA"], ones, True, tsign=t)
        hnB_f, mwB_f = K.filip_prep(dv["hB"], ones, True)
        ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]
        for i in range(3 + args.steps):
            if i >= 3:
                ev[i - 3][0].record()
            K.filip_sim_fwd(hnA_f, mwA_f, L, hnB_f, mwB_f, L, t, want_arg=True, unit_norm=True)
            if i >= 3:
                ev[i - 3][1].record()
        torch.cuda.synchronize()
        full_ms = statistics.median(a.elapsed_time(b) for a, b in ev)

    clip_bg, filip_bg = clip_bl * world, filip_bl * world
    pairs = (clip_bg if do_clip else 0) + (filip_bg if do_filip else 0)

    # algorithmic work of the dominant kernel (FILIP score kernel), this rank's launch:
    # 2 * d * (valid A tokens of the rank) * (valid B tokens of the global batch)
    if do_filip:
        nA = int(cpu["lA"].sum())
        nB_local = torch.tensor([float(cpu["lB"].sum())], device=device, dtype=torch.float64)
        if multi:
            dist.all_reduce(nB_local)
        flops = 2.0 * D * nA * nB_local.item()
        achieved = flops / (sim_ms * 1e-3) / 1e12
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "filip_sim_traffic.json")
        if world == 1 and os.path.exists(tpath):
            with open(tpath) as f:
                traffic = json.load(f).get("hbm_bytes_per_launch")
        roof = {"bound": "mfma", "kernel": "filip_sim_stream_kernel", "achieved": achieved, "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s",
                "frac": achieved / PEAK_BF16_TFLOPS, "traffic": traffic, "kernel_ms": sim_ms, "algorithmic_flops_per_launch": flops}
    else:
        # CLIP alone: 16*B*d algorithmic bytes per step over the whole step time (HBM roof)
        byts = 16.0 * clip_bl * D + (8.0 * clip_bg * D if world > 1 else 0.0)
        ach = byts / (t_clip / args.steps) / 1e9
        roof = {"bound": "hbm", "kernel": "clip step (all kernels)", "achieved": ach, "peak": PEAK_HBM_GBS, "unit": "GB/s",
                "frac": ach / PEAK_HBM_GBS, "traffic": None}

    if rank == 0:
        workloads = {}
        if do_clip:
            ms = t_clip / args.steps * 1e3
            flop = 6.0 * clip_bl * clip_bg * D
            byts = 16.0 * clip_bl * D + (8.0 * clip_bg * D if world > 1 else 0.0)
            workloads["clip"] = {
                "global_batch": clip_bg, "per_gpu_batch": clip_bl, "d": D, "dtype": "f32", "ms_per_step": ms,
                "pairs_per_s": clip_bg / (ms * 1e-3),
                "algorithmic_bytes_per_gpu": byts, "algorithmic_flops_per_gpu": flop,
                "hbm_GBs": byts / (ms * 1e-3) / 1e9, "hbm_frac": byts / (ms * 1e-3) / 1e9 / PEAK_HBM_GBS,
                "f32_mfma_TFLOPs": flop / (ms * 1e-3) / 1e12, "f32_mfma_frac": flop / (ms * 1e-3) / 1e12 / PEAK_F32_MFMA_TFLOPS,
            }
        if do_filip:
            ms = t_filip / args.steps * 1e3
            workloads["filip"] = {
                "global_batch": filip_bg, "per_gpu_batch": filip_bl, "L": L, "d": D, "dtype": "bf16 operands, f32 accumulate",
                "ms_per_step": ms, "pairs_per_s": filip_bg / (ms * 1e-3), "scored_pairs_per_s": filip_bl * filip_bg * world / (ms * 1e-3),
                "sim_kernel_ms": sim_ms, "sim_kernel_TFLOPs": roof["achieved"], "sim_kernel_mfma_frac": roof["frac"],
            }
            # the whole step against the matrix peak: the algorithmic flops of the scores (the backward's routed row
            # additions are 1.5 % of them and not counted) over the step time, so that what the backward, the
            # preparation and the losses cost is visible beside the score kernel's own fraction
            workloads["filip"]["step_mfma_frac"] = roof["algorithmic_flops_per_launch"] / (ms * 1e-3) / 1e12 / PEAK_BF16_TFLOPS
            if bwd_ms is not None:
                # marks of mutual choices, inversions, both gathers and the prep backward (one C call)
                workloads["filip"]["backward_ms"] = bwd_ms
            if full_ms is not None:
                ff = 2.0 * D * (filip_bl * L) ** 2
                workloads["filip_full_length_scores"] = {
                    "note": "score kernel alone, every sequence 256 valid tokens (not part of `value`)",
                    "sim_kernel_ms": full_ms, "sim_kernel_TFLOPs": ff / (full_ms * 1e-3) / 1e12,
                    "sim_kernel_mfma_frac": ff / (full_ms * 1e-3) / 1e12 / PEAK_BF16_TFLOPS,
                }
        if not multi and do_filip and not args.no_rank_shapes:
            # context for `roofline.frac` (which stays against the nominal peak): what the library's GEMM sustains here
            lib = library_gemm_tflops(device)
            roof["library_gemm"] = {"TFLOPs": lib, "what": "torch.matmul bf16 8192^3, random operands, same GPU, same run",
                                    "achieved_over_it": roof["achieved"] / lib}
            if "filip_full_length_scores" in workloads:
                workloads["filip_full_length_scores"]["over_library_gemm"] = workloads["filip_full_length_scores"]["sim_kernel_TFLOPs"] / lib
        if not multi and not args.no_rank_shapes and args.workload == "both":
            workloads.update(rank_shape_records(K, device))
            workloads.update(head_record(K, device))
            workloads.update(long_chain_record(K, device))
        out = {
            "metric": "protein_pairs_per_sec", "value": pairs / (total / args.steps), "unit": "pairs/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "settle_steps": SETTLE_STEPS, "ms_per_step": total / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "bf16" if do_filip else "f32", "data": "synthetic",
            "config": {
                "workload": {"both": f"clip_b{clip_bg}_d{D}+filip_b{filip_bg}_l{L}_d{D}", "clip": f"clip_b{clip_bg}_d{D}",
                             "filip": f"filip_b{filip_bg}_l{L}_d{D}"}[args.workload],
                "clip_global_batch": clip_bg, "filip_global_batch": filip_bg, "seq_len": L, "d": D,
                "parallelism": f"dp{world}", "step": "fwd+bwd from raw head outputs to their gradients, both workloads back to back",
            },
            "roofline": roof,
            "workloads": workloads,
            "device_allocations_in_timed_steps": step_allocs,
        }
        if rehearse:
            out["note"] = "rehearsal of the multi-rank code on one GPU (--rehearse-dist); not a benchmark result"
        if not multi and not args.no_cpu_baseline and args.workload == "both":
            out["cpu_baseline"] = cpu_baseline(cpu, clip_bl, filip_bl)
        sys.stdout.flush()
        os.write(real_stdout, (json.dumps(out) + "\n").encode())
    if multi:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
