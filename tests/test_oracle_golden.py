"""Pins the CPU oracle (oracle/pclip_oracle.py) to the golden vectors produced by the
reference's own functions (oracle/gen_golden.py).  CPU only."""
import torch
import pytest

from oracle import pclip_oracle as O
from conftest import load_flat, head_case_params


def close(a, b, rtol, atol):
    torch.testing.assert_close(a, b.to(a.dtype), rtol=rtol, atol=atol)


def test_clip_small_cases(clip_cases):
    assert len(clip_cases) == 6
    for key, c in clip_cases.items():
        t = c["t"]
        x = O.l2norm_scale(c["e_pep"], t)
        y = O.l2norm_scale(c["e_rec"], t)
        close(x, c["x"], 1e-6, 1e-7)
        close(y, c["y"], 1e-6, 1e-7)
        close(O.clip_logits(x, y), c["logits"], 1e-5, 1e-6)
        close(O.clip_contrastive_loss(x, y.t()), c["loss"], 1e-6, 1e-6)
        xs = list(torch.chunk(x, 2)); ys = list(torch.chunk(y, 2))
        close(O.clip_contrastive_loss_gc(xs, ys), c["loss_gc2"], 1e-6, 1e-6)
        close(O.clip_contrastive_loss_gc(list(torch.chunk(x, 4)), list(torch.chunk(y, 4))), c["loss_gc4"], 1e-6, 1e-6)
        close(O.clip_get_logits(xs, ys), c["exp_logits"], 1e-5, 1e-6)
        # the reference's list variant equals its plain variant
        close(c["loss_gc2"], c["loss"], 1e-6, 1e-6)
        for step in (O.clip_step, O.clip_step_analytic):
            loss, dp, dr, dt = step(c["e_pep"], c["e_rec"], t)
            close(loss, c["loss"], 1e-6, 1e-6)
            close(dp, c["d_e_pep"], 1e-4, 2e-7)
            close(dr, c["d_e_rec"], 1e-4, 2e-7)
            close(dt, c["d_t"], 1e-4, 1e-6)


def test_clip_scalars_at_benchmark_sizes():
    g = load_flat("clip_scalars.npz")
    for B in (256, 1024):
        gen = torch.Generator().manual_seed(0)
        e_pep = torch.randn(B, 128, generator=gen)
        e_rec = torch.randn(B, 128, generator=gen)
        t = torch.tensor(1.0)
        x = O.l2norm_scale(e_pep, t)
        close(O.clip_contrastive_loss(x, O.l2norm_scale(e_rec, t).t()), g[f"loss_B{B}"], 1e-6, 1e-6)
        e_rec2 = e_pep + 0.5 * torch.randn(B, 128, generator=gen)
        close(O.clip_contrastive_loss(x, O.l2norm_scale(e_rec2, t).t()), g[f"loss_peaked_B{B}"], 1e-6, 1e-6)


def test_clip_uniform_logits_give_log_B():
    S = torch.zeros(16, 16)
    close(O.clip_loss_from_logits(S), torch.log(torch.tensor(16.0)), 1e-6, 1e-6)


def test_filip_small_cases(filip_cases):
    assert len(filip_cases) == 7
    for key, c in filip_cases.items():
        t = c["t"]
        pre = key.endswith("bf16n")
        if pre:
            sA, sB = O.filip_similarity(c["hA"], c["hB"], c["maskA"], c["maskB"], t, block=3)
        else:
            sA, sB = O.filip_similarity(O.filip_normalise(c["hA"]), O.filip_normalise(c["hB"]),
                                        c["maskA"], c["maskB"], t, block=3)
        close(sA, c["sim_A"], 1e-5, 1e-6)
        close(sB, c["sim_B"], 1e-5, 1e-6)
        close(O.filip_loss(sA, sB), c["loss"], 1e-5, 1e-6)
        close(O.row_ce(c["sim_A"]), c["loss_A"], 1e-5, 1e-6)
        close(O.row_ce(c["sim_B"], use_dcl=True), c["loss_B_dcl"], 1e-5, 1e-6)
        if not pre:
            loss, s1, s2, dA, dB, dt = O.filip_step(c["hA"], c["hB"], c["maskA"], c["maskB"], t, block=4)
            close(loss, c["loss"], 1e-5, 1e-6)
            close(dA, c["d_hA"], 1e-3, 1e-7)
            close(dB, c["d_hB"], 1e-3, 1e-7)
            close(dt, c["d_t"], 1e-4, 1e-6)


def test_filip_block_decomposition_matches_one_shot(filip_cases):
    c = filip_cases["s2_B8_TA48_TB40_d128_raw"]
    nA, nB = O.filip_normalise(c["hA"]), O.filip_normalise(c["hB"])
    a1, b1 = O.filip_similarity(nA, nB, c["maskA"], c["maskB"], c["t"], block=8)
    a2, b2 = O.filip_similarity(nA, nB, c["maskA"], c["maskB"], c["t"], block=3)
    assert torch.equal(a1, a2) and torch.equal(b1, b2)


def test_filip_benchmark_size_rows():
    """A slice of the B=256, L=256 reference scores (rows 0..7) from regenerated inputs."""
    g = load_flat("filip_b256_l256.npz")
    B, L, d = 256, 256, 128
    gen = torch.Generator().manual_seed(0)
    hA = O.bf16_round(torch.randn(B, L, d, generator=gen))
    hB = O.bf16_round(torch.randn(B, L, d, generator=gen))
    lA = torch.randint(128, 257, (B,), generator=gen)
    lB = torch.randint(128, 257, (B,), generator=gen)
    assert torch.equal(lA, g["lenA"]) and torch.equal(lB, g["lenB"])
    mA, mB = O.right_padded_masks(lA, L), O.right_padded_masks(lB, L)
    sA, sB = O.filip_similarity(O.filip_normalise(hA[:8]), O.filip_normalise(hB[:32]), mA[:8], mB[:32],
                                torch.tensor(1.0), block=8)
    close(sA, g["sim_A"][:8, :32], 1e-5, 1e-6)
    close(sB, g["sim_B"][:8, :32], 1e-5, 1e-6)
    close(O.filip_loss(g["sim_A"], g["sim_B"]), g["loss"], 1e-6, 1e-6)


def test_filip_benchmark_size_rows_bf16_exact_operands():
    """The second benchmark-size fixture: tokens normalised, THEN rounded to bf16, fed to the reference's score
    function as they are (what a bf16 matrix kernel computes on exactly).  A slice from regenerated inputs."""
    g = load_flat("filip_b256_l256_bf16n.npz")
    B, L, d = 256, 256, 128
    gen = torch.Generator().manual_seed(0)
    hA = O.bf16_round(torch.randn(B, L, d, generator=gen))
    hB = O.bf16_round(torch.randn(B, L, d, generator=gen))
    lA = torch.randint(128, 257, (B,), generator=gen)
    lB = torch.randint(128, 257, (B,), generator=gen)
    assert torch.equal(lA, g["lenA"]) and torch.equal(lB, g["lenB"])
    mA, mB = O.right_padded_masks(lA, L), O.right_padded_masks(lB, L)
    nA = O.bf16_round(hA[100:108] / hA[100:108].norm(dim=-1, keepdim=True))   # the generator's expression (torch's norm)
    nB = O.bf16_round(hB[64:96] / hB[64:96].norm(dim=-1, keepdim=True))
    sA, sB = O.filip_similarity(nA, nB, mA[100:108], mB[64:96], torch.tensor(1.0), block=8)
    close(sA, g["sim_A"][100:108, 64:96], 1e-6, 1e-6)
    close(sB, g["sim_B"][100:108, 64:96], 1e-6, 1e-6)
    close(O.filip_loss(g["sim_A"], g["sim_B"]), g["loss"], 1e-6, 1e-6)
    # rounding the unit vectors to bf16 moves a score by up to a few 1e-4: the two fixtures differ by that much
    g32 = load_flat("filip_b256_l256.npz")
    assert 1e-5 < (g["sim_A"] - g32["sim_A"]).abs().max() < 2e-3


def test_filip_exact_ties_split_evenly_like_the_reference():
    """Duplicated tokens (exact ties of the max): the oracle's gradient equals the reference's
    even split among the tied tokens, element by element."""
    c = load_flat("filip_ties.npz")
    a = c["hA"].clone().requires_grad_(True)
    b = c["hB"].clone().requires_grad_(True)
    t = c["t"].clone().requires_grad_(True)
    sA, sB = O.filip_similarity_block(a, b, c["maskA"], c["maskB"], t)
    close(sA, c["sim_A"], 1e-6, 1e-7)
    close(sB, c["sim_B"], 1e-6, 1e-7)
    O.filip_loss(sA, sB).backward()
    close(a.grad, c["d_hA"], 1e-5, 1e-9)
    close(b.grad, c["d_hB"], 1e-5, 1e-9)
    close(t.grad, c["d_t"], 1e-5, 1e-7)
    # the fixture does contain ties: twins receive equal, nonzero shares
    twins = (c["groupB"] != torch.arange(c["groupB"].numel())).nonzero().flatten()
    assert len(twins) == 5
    for tw in twins.tolist():
        rep = int(c["groupB"][tw])
        assert torch.equal(c["d_hB"][:, tw], c["d_hB"][:, rep]) and c["d_hB"][:, rep].abs().sum() > 0


def test_topk_eval_host_logic_against_the_reference_ranks():
    """``eval_gc_allrec_onepep`` (training_utils.py:174-228 of the reference) on the tiny model of
    the loop traces, compute through the CPU stand-in provider: same rank list for every k."""
    import json
    import os
    import cpu_provider as P
    from torch.utils.data import DataLoader
    from protein_clip_amd import ops, synthetic as S
    from protein_clip_amd.modules import models, training_utils
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    with open(os.path.join(root, "tests", "golden", "topk_trace.json")) as f:
        want = json.load(f)
    tok = S.make_tokenizer()
    data = S.PairDataset(8, pep_len=(5, 12), rec_len=(10, 30), seed=7)
    torch.manual_seed(want["seed"])
    model = models.ExtendedCLIP(32, 16, 2, 2, 0.0, S.make_esm(32, 2, 4))
    loader = DataLoader(data, batch_size=want["batch_size"], shuffle=False, drop_last=True)
    real = ops._hip_kernels
    ops._hip_kernels = P
    try:
        for k in range(8):
            got = training_utils.eval_gc_allrec_onepep(model, loader, "cpu", tok, agg_batches=want["agg_batches"], k=k)
            assert got == want["ranks"][str(k)], (k, got, want["ranks"][str(k)], want["diag"][str(k)])
    finally:
        ops._hip_kernels = real


def _train_gc_trace():
    import json
    import os
    import numpy as np
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    with open(os.path.join(root, "tests", "golden", "train_gc_trace.json")) as f:
        want = json.load(f)
    z = np.load(os.path.join(root, "tests", "golden", "train_gc_params.npz"), allow_pickle=False)
    return want, {k: torch.from_numpy(np.asarray(z[k])) for k in z.files}


def run_train_gc_against_trace(device, loss_tol, param_rel):
    """Shared by the CPU (stand-in provider) and GPU tests: ``train_gc`` on the pairs, model and optimizer of
    ``tests/golden/train_gc_trace.json``, which holds what the REFERENCE's ``train_gc``
    (training_utils.py:92-155) returned and left behind: the loss of every big batch, the epoch means,
    and temperature and head parameters after each epoch (20 pairs, sub-batches of 4, two per big batch,
    so the fifth sub-batch of an epoch is encoded and dropped)."""
    from torch.utils.data import DataLoader
    from protein_clip_amd import synthetic as S
    from protein_clip_amd.modules import models, training_utils

    class Pairs(torch.utils.data.Dataset):
        def __init__(self, pep, rec):
            self.pep, self.rec = pep, rec

        def __len__(self):
            return len(self.pep)

        def __getitem__(self, i):
            return self.pep[i], self.rec[i]

    want, params = _train_gc_trace()
    tok = S.make_tokenizer()
    torch.manual_seed(want["seed"])
    esm = S.make_esm(want["esm"]["hidden"], want["esm"]["layers"], want["esm"]["heads"])
    for prm in esm.parameters():
        prm.requires_grad = False
    h = want["head"]
    model = models.ExtendedCLIP(h["input_dim"], h["embedding_dim"], h["h1"], h["h2"], h["dropout"], esm).to(device)
    start = {n: p.detach().cpu().clone() for n, p in model.named_parameters() if p.requires_grad}
    loader = DataLoader(Pairs(want["peptides"], want["receptors"]), batch_size=want["batch_size"], shuffle=False, drop_last=True)
    opt = torch.optim.Adam(model.parameters(), lr=want["lr"])
    scaler = torch.amp.GradScaler("cuda", enabled=False)
    seen = []
    inner = training_utils._contrastive_loss_gc

    def recording(x, y, group=None):
        loss = inner(x, y, group=group)
        seen.append(float(loss.item()))
        return loss
    training_utils._contrastive_loss_gc = recording
    try:
        for ep, w in enumerate(want["epochs"]):
            seen.clear()
            mean = training_utils.train_gc(model, loader, tok, opt, scaler, device, accumulated_batches=want["accumulated_batches"])
            assert len(seen) == len(w["big_batch_losses"]) == 2      # the trailing fifth sub-batch formed no batch
            for got, ref in zip(seen, w["big_batch_losses"]):
                assert abs(got - ref) < loss_tol, (ep, seen, w)
            assert abs(mean - w["mean"]) < loss_tol
            assert abs(model.temperature.item() - w["temperature"]) < loss_tol
            # parameters: the distance travelled from the start is what the steps produced; compare that
            num = den = 0.0
            for n, p in model.named_parameters():
                if p.requires_grad:
                    ref = params[f"epoch{ep}/{n}"]
                    num += float((p.detach().cpu() - ref).pow(2).sum())
                    den += float((ref - start[n]).pow(2).sum())
            assert den > 0 and (num / den) ** 0.5 < param_rel, (ep, (num / den) ** 0.5)
    finally:
        training_utils._contrastive_loss_gc = inner


def test_train_gc_host_logic_against_the_reference_trace():
    """The big-batch trainer's host logic (grouping of sub-batches, the dropped trailing sub-batch, replay, order of
    step / update / zero_grad, epoch mean) against the reference's own ``train_gc``; compute through the CPU
    stand-in provider."""
    import cpu_provider as P
    from protein_clip_amd import ops
    want, _ = _train_gc_trace()
    for w in want["epochs"]:
        assert abs(sum(w["big_batch_losses"]) / len(w["big_batch_losses"]) - w["mean"]) < 1e-6   # training_utils.py:155
    real = ops._hip_kernels
    ops._hip_kernels = P
    try:
        run_train_gc_against_trace("cpu", 2e-5, 1e-4)   # measured: passes at 1e-5
    finally:
        ops._hip_kernels = real


def test_head_tail_cases(head_cases):
    """Pooling, embedding_ffn, normalise and scale of the reference's Encoder (forward in f32, gradients
    checked in f64 as well): values, and every gradient autograd returned."""
    assert len(head_cases) == 4
    for key, c in head_cases.items():
        params, grads, keep, scale = head_case_params(c)
        assert len(params) == 4 * (int(key.split("_h")[1].split("_")[0]) - 1) + 2
        for dt_, rtol, atol in ((torch.float32, 2e-4, 2e-6), (torch.float64, 2e-4, 2e-6)):
            cast = lambda v: v.to(dt_)
            out, pooled, d_tok, d_pool, d_prm, d_t = O.head_step(cast(c["tokens"]), c["mask"], [cast(p) for p in params], keep, scale, 1e-5,
                                                                 cast(c["t"]), cast(c["d_out"]))
            close(pooled.float(), c["pooled"], 1e-5, 1e-6)
            close(out.float(), c["out"], 1e-4, 1e-6)
            close(d_pool.float(), c["d_pooled"], rtol, atol)
            close(d_tok.float(), c["d_tokens"], rtol, atol)
            close(d_t.float(), c["d_t"], rtol, 1e-5)
            for g, want in zip(d_prm, grads):
                close(g.float(), want, rtol, 2e-5)
