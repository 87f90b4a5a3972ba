"""Generate the golden vectors under ``tests/golden/`` by running the REFERENCE's own
functions on seeded inputs (TEST INFRASTRUCTURE ONLY; run in the authoring container, where
the read-only reference checkout is mounted — it does not exist on the GPU box).

    PYTHONDONTWRITEBYTECODE=1 python oracle/gen_golden.py [/root/reference]

What is imported from the reference: ``modules.models`` (ExtendedFILIP static methods, Encoder),
``modules.training_utils_filip`` and ``modules.training_utils``.  The last one imports the
third-party ``grad_cache`` package, which is not vendored and not installable offline; a
pass-through stand-in for its two decorators (list arguments concatenated on dim 0 /
identity) is registered first.  That stand-in is ours and contributes no arithmetic.

Only inputs and outputs are stored (``.npz``, float32/uint8/int64 arrays): no reference
source text goes into the repository.
"""
from __future__ import annotations

import os
import sys
import types

import numpy as np
import torch
import torch.nn.functional as F

REF = next((a for a in sys.argv[1:] if not a.startswith("--")), "/root/reference")
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden")


def _install_grad_cache_stand_in():
    def cat_input_tensor(fn):
        def wrapped(*args, **kw):
            args = [torch.cat(list(a), dim=0) if isinstance(a, (list, tuple)) else a for a in args]
            return fn(*args, **kw)
        return wrapped

    def cached(fn):
        # what the package's decorator hands back for one tensor: the representation computed
        # without a graph, as a leaf, and a closure that replays the forward for the backward
        def wrapped(*args, **kw):
            with torch.no_grad():
                rep = fn(*args, **kw)
            leaf = rep.detach().requires_grad_(True)

            def replay(cached_rep):
                again = fn(*args, **kw)
                torch.autograd.backward(again, grad_tensors=cached_rep.grad)
            return leaf, replay
        return wrapped

    pkg = types.ModuleType("grad_cache")
    sub = types.ModuleType("grad_cache.functional")
    sub.cat_input_tensor = cat_input_tensor
    sub.cached = cached
    pkg.functional = sub
    sys.modules["grad_cache"] = pkg
    sys.modules["grad_cache.functional"] = sub


def _np(t):
    return t.detach().cpu().numpy()


def gen_clip(tu):
    cases = {}
    for seed, (B, d), t0, peaked in [
        (0, (4, 8), 1.0, False),
        (1, (16, 128), 0.0, False),
        (2, (32, 128), 2.0, False),
        (3, (32, 128), 1.0, True),
        (4, (37, 128), 1.0, False),      # ragged batch (not a multiple of any tile)
        (5, (64, 96), 0.5, False),       # d not a power of two
    ]:
        g = torch.Generator().manual_seed(seed)
        e_pep = torch.randn(B, d, generator=g)
        e_rec = e_pep + 0.1 * torch.randn(B, d, generator=g) if peaked else torch.randn(B, d, generator=g)
        ep = e_pep.clone().requires_grad_(True)
        er = e_rec.clone().requires_grad_(True)
        t = torch.tensor(t0, requires_grad=True)
        # modules/models.py:50-51 are two torch calls inside Encoder.forward (which needs
        # the ESM trunk); they are issued here verbatim on the head output.
        x = F.normalize(ep, dim=-1) * torch.exp(t / 2)
        y = F.normalize(er, dim=-1) * torch.exp(t / 2)
        loss = tu._contrastive_loss(x, y.t())
        loss.backward()
        xs2 = list(torch.chunk(x.detach(), 2, 0)); ys2 = list(torch.chunk(y.detach(), 2, 0))
        xs4 = list(torch.chunk(x.detach(), 4, 0)); ys4 = list(torch.chunk(y.detach(), 4, 0))
        key = f"s{seed}_B{B}_d{d}"
        cases[key] = dict(
            e_pep=_np(e_pep), e_rec=_np(e_rec), t=np.float32(t0),
            x=_np(x), y=_np(y), loss=_np(loss), logits=_np(x.detach() @ y.detach().t()),
            exp_logits=_np(tu.get_logits(xs2, ys2)),
            loss_gc2=_np(tu._contrastive_loss_gc(xs2, ys2)),
            loss_gc4=_np(tu._contrastive_loss_gc(xs4, ys4)),
            d_e_pep=_np(ep.grad), d_e_rec=_np(er.grad), d_t=_np(t.grad),
        )
    flat = {f"{k}/{f}": v for k, c in cases.items() for f, v in c.items()}
    np.savez_compressed(os.path.join(OUT, "clip_small.npz"), **flat)

    # loss-only scalars at the benchmark sizes (inputs are regenerated from the seed)
    scal = {}
    for B in (256, 1024):
        g = torch.Generator().manual_seed(0)
        e_pep = torch.randn(B, 128, generator=g)
        e_rec = torch.randn(B, 128, generator=g)
        t = torch.tensor(1.0)
        x = F.normalize(e_pep, dim=-1) * torch.exp(t / 2)
        y = F.normalize(e_rec, dim=-1) * torch.exp(t / 2)
        scal[f"loss_B{B}"] = _np(tu._contrastive_loss(x, y.t()))
        e_rec2 = e_pep + 0.5 * torch.randn(B, 128, generator=g)
        y2 = F.normalize(e_rec2, dim=-1) * torch.exp(t / 2)
        scal[f"loss_peaked_B{B}"] = _np(tu._contrastive_loss(x, y2.t()))
    np.savez_compressed(os.path.join(OUT, "clip_scalars.npz"), **scal)


def _filip_inputs(seed, B, TA, TB, d, full=False, lo=3):
    g = torch.Generator().manual_seed(seed)
    hA = torch.randn(B, TA, d, generator=g)
    hB = torch.randn(B, TB, d, generator=g)
    if full:
        lA = torch.full((B,), TA); lB = torch.full((B,), TB)
    else:
        lA = torch.randint(min(lo, TA), TA + 1, (B,), generator=g)
        lB = torch.randint(min(lo, TB), TB + 1, (B,), generator=g)
    mA = (torch.arange(TA)[None] < lA[:, None]).long()
    mB = (torch.arange(TB)[None] < lB[:, None]).long()
    return hA, hB, mA, mB


def gen_filip(models, tuf):
    FIL = models.ExtendedFILIP
    cases = {}
    for seed, (B, TA, TB, d), t0, kind in [
        (0, (3, 5, 7, 8), 1.0, "raw"),
        (1, (6, 9, 11, 16), 0.7, "raw"),
        (2, (8, 48, 40, 128), 1.0, "raw"),
        (3, (8, 48, 40, 128), 0.7, "bf16n"),   # operands already normalised and bf16-exact
        (4, (5, 70, 33, 128), 1.3, "bf16n"),
        (5, (4, 64, 64, 128), 1.0, "full"),    # no padding at all
        (6, (6, 37, 29, 64), 1.0, "holes"),    # masks with interior holes (not right-padded)
    ]:
        hA, hB, mA, mB = _filip_inputs(seed, B, TA, TB, d, full=(kind == "full"))
        if kind == "holes":
            g = torch.Generator().manual_seed(100 + seed)
            mA = (torch.rand(B, TA, generator=g) > 0.3).long(); mA[:, 0] = 1
            mB = (torch.rand(B, TB, generator=g) > 0.3).long(); mB[:, 0] = 1
        key = f"s{seed}_B{B}_TA{TA}_TB{TB}_d{d}_{kind}"
        t = torch.tensor(t0, requires_grad=True)
        if kind == "bf16n":
            # pre-normalised, bf16-representable operands fed straight to the score function
            nA = (hA / hA.norm(dim=-1, keepdim=True)).to(torch.bfloat16).float()
            nB = (hB / hB.norm(dim=-1, keepdim=True)).to(torch.bfloat16).float()
            a = nA.clone().requires_grad_(True); b = nB.clone().requires_grad_(True)
            sA, sB = FIL._filip_similarity_score(a, b, mA, mB, t)
            store_A, store_B = nA, nB
        else:
            a = hA.clone().requires_grad_(True); b = hB.clone().requires_grad_(True)
            # modules/models.py:238-239 (two lines of ExtendedFILIP.forward, which needs the trunk)
            na = a / a.norm(dim=-1, keepdim=True)
            nb = b / b.norm(dim=-1, keepdim=True)
            sA, sB = FIL._filip_similarity_score(na, nb, mA, mB, t)
            store_A, store_B = hA, hB
        loss = tuf._compute_loss(sA, sB)
        loss.backward()
        cases[key] = dict(
            hA=_np(store_A), hB=_np(store_B), maskA=_np(mA).astype(np.uint8), maskB=_np(mB).astype(np.uint8),
            t=np.float32(t0), sim_A=_np(sA), sim_B=_np(sB), loss=_np(loss),
            loss_A=_np(tuf._contrastive_loss(sA.detach())), loss_B_dcl=_np(tuf._contrastive_loss(sB.detach(), use_dcl=True)),
            d_hA=_np(a.grad), d_hB=_np(b.grad), d_t=_np(t.grad),
        )
    flat = {f"{k}/{f}": v for k, c in cases.items() for f, v in c.items()}
    np.savez_compressed(os.path.join(OUT, "filip_small.npz"), **flat)

    # benchmark-size instance: B=256, L=256, d=128, bf16-rounded raw inputs, lengths in [128,256];
    # the reference score function is run on 16x16 pair blocks (exact: a pair's scores depend
    # on that pair alone).  Only sim_A/sim_B/loss are kept (inputs are regenerated from the seed).
    B, L, d = 256, 256, 128
    g = torch.Generator().manual_seed(0)
    hA = torch.randn(B, L, d, generator=g).to(torch.bfloat16).float()
    hB = torch.randn(B, L, d, generator=g).to(torch.bfloat16).float()
    lA = torch.randint(128, 257, (B,), generator=g)
    lB = torch.randint(128, 257, (B,), generator=g)
    mA = (torch.arange(L)[None] < lA[:, None]).long()
    mB = (torch.arange(L)[None] < lB[:, None]).long()
    nA = hA / hA.norm(dim=-1, keepdim=True)
    nB = hB / hB.norm(dim=-1, keepdim=True)
    t = torch.tensor(1.0)
    sA = torch.empty(B, B); sB = torch.empty(B, B)
    blk = 16
    with torch.no_grad():
        for i0 in range(0, B, blk):
            for j0 in range(0, B, blk):
                a_, b_ = FIL._filip_similarity_score(nA[i0:i0+blk], nB[j0:j0+blk], mA[i0:i0+blk], mB[j0:j0+blk], t)
                sA[i0:i0+blk, j0:j0+blk] = a_; sB[i0:i0+blk, j0:j0+blk] = b_
            print("filip B256 block row", i0, flush=True)
        loss = tuf._compute_loss(sA, sB)
    np.savez_compressed(os.path.join(OUT, "filip_b256_l256.npz"), sim_A=_np(sA), sim_B=_np(sB), loss=_np(loss),
                        lenA=_np(lA), lenB=_np(lB))


def gen_filip_b256_bf16n(models, tuf):
    """Benchmark-size instance on operands a bf16 matrix kernel sees EXACTLY: the tokens are normalised in f32 and
    then rounded to bf16 (what ``pclip_filip_prep`` writes), and those rounded unit vectors go straight to the
    reference's ``_filip_similarity_score`` (models.py:292-356).  All 2 x 65,536 scores and the loss are stored, so
    the whole matrix can be compared at the tolerance of the small cases.  Inputs are regenerated from the seed."""
    FIL = models.ExtendedFILIP
    B, L, d = 256, 256, 128
    g = torch.Generator().manual_seed(0)
    hA = torch.randn(B, L, d, generator=g).to(torch.bfloat16).float()
    hB = torch.randn(B, L, d, generator=g).to(torch.bfloat16).float()
    lA = torch.randint(128, 257, (B,), generator=g)
    lB = torch.randint(128, 257, (B,), generator=g)
    mA = (torch.arange(L)[None] < lA[:, None]).long()
    mB = (torch.arange(L)[None] < lB[:, None]).long()
    nA = (hA / hA.norm(dim=-1, keepdim=True)).to(torch.bfloat16).float()
    nB = (hB / hB.norm(dim=-1, keepdim=True)).to(torch.bfloat16).float()
    t = torch.tensor(1.0)
    sA = torch.empty(B, B); sB = torch.empty(B, B)
    blk = 16
    with torch.no_grad():
        for i0 in range(0, B, blk):
            for j0 in range(0, B, blk):
                a_, b_ = FIL._filip_similarity_score(nA[i0:i0+blk], nB[j0:j0+blk], mA[i0:i0+blk], mB[j0:j0+blk], t)
                sA[i0:i0+blk, j0:j0+blk] = a_; sB[i0:i0+blk, j0:j0+blk] = b_
            print("filip B256 bf16n block row", i0, flush=True)
        loss = tuf._compute_loss(sA, sB)
    np.savez_compressed(os.path.join(OUT, "filip_b256_l256_bf16n.npz"), sim_A=_np(sA), sim_B=_np(sB), loss=_np(loss),
                        lenA=_np(lA), lenB=_np(lB))


def gen_loops(models, tu, tuf):
    """Two epochs of the reference's own train/evaluate loops on a tiny randomly initialised
    ESM-2 (hidden 32, 2 layers) and 8 synthetic sequence pairs: pins the loop order
    (zero_grad / backward / step), the loss averaging and the state-dict key names."""
    import json
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
    from protein_clip_amd import synthetic as S
    from torch.utils.data import DataLoader
    tok = S.make_tokenizer()
    data = S.PairDataset(8, pep_len=(5, 12), rec_len=(10, 30), seed=7)
    out = {"peptides": data.pep, "receptors": data.rec, "batch_size": 4, "esm": {"hidden": 32, "layers": 2, "heads": 4},
           "head": {"input_dim": 32, "embedding_dim": 16, "h1": 2, "h2": 2, "dropout": 0.0}, "lr": 1e-3, "seed": 42}
    for name, cls, mod in (("clip", models.ExtendedCLIP, tu), ("filip", models.ExtendedFILIP, tuf)):
        torch.manual_seed(42)
        esm = S.make_esm(32, 2, 4)
        for prm in esm.parameters():
            prm.requires_grad = False
        model = cls(32, 16, 2, 2, 0.0, esm)
        loader = DataLoader(data, batch_size=4, shuffle=False, drop_last=True)
        opt = torch.optim.Adam(model.parameters(), lr=1e-3)
        trace = {"state_dict_keys": list(model.state_dict().keys()), "epochs": []}
        for _ in range(2):
            tr = mod.train(model, loader, opt, tok, "cpu")
            ev = mod.evaluate(model, loader, tok, "cpu")
            trace["epochs"].append({"train": float(tr), "eval": float(ev), "temperature": float(model.temperature.item())})
        out[name] = trace
    with open(os.path.join(OUT, "loops_trace.json"), "w") as f:
        json.dump(out, f, indent=1)


def gen_ties(models, tuf):
    """Exact ties: duplicated tokens on both sides (operands normalised and bf16-exact).  The
    reference's ``amax`` backward splits a gradient evenly among tied maxima; a kernel that routes
    it to ONE of the tied tokens agrees after summing over each group of identical tokens (and
    everywhere for the other side, whose gradient is the same row whichever twin is picked)."""
    FIL = models.ExtendedFILIP
    B, TA, TB, d = 5, 40, 37, 128
    g = torch.Generator().manual_seed(77)
    hA = torch.randn(B, TA, d, generator=g)
    hB = torch.randn(B, TB, d, generator=g)
    groupA = torch.arange(TA); groupB = torch.arange(TB)
    for a, b in [(3, 2), (9, 2), (33, 32)]:        # token a is a copy of token b (one triple, one across a 32-tile edge)
        hA[:, a] = hA[:, b]; groupA[a] = b
    for a, b in [(1, 0), (5, 4), (6, 4), (7, 4), (35, 20)]:
        hB[:, a] = hB[:, b]; groupB[a] = b
    lA = torch.randint(34, TA + 1, (B,), generator=g)
    lB = torch.randint(36, TB + 1, (B,), generator=g)
    mA = (torch.arange(TA)[None] < lA[:, None]).long()
    mB = (torch.arange(TB)[None] < lB[:, None]).long()
    nA = (hA / hA.norm(dim=-1, keepdim=True)).to(torch.bfloat16).float()
    nB = (hB / hB.norm(dim=-1, keepdim=True)).to(torch.bfloat16).float()
    a = nA.clone().requires_grad_(True); b = nB.clone().requires_grad_(True)
    t = torch.tensor(0.8, requires_grad=True)
    sA, sB = FIL._filip_similarity_score(a, b, mA, mB, t)
    loss = tuf._compute_loss(sA, sB)
    loss.backward()
    np.savez_compressed(os.path.join(OUT, "filip_ties.npz"), hA=_np(nA), hB=_np(nB), maskA=_np(mA).astype(np.uint8),
                        maskB=_np(mB).astype(np.uint8), t=np.float32(0.8), sim_A=_np(sA), sim_B=_np(sB), loss=_np(loss),
                        d_hA=_np(a.grad), d_hB=_np(b.grad), d_t=_np(t.grad), groupA=_np(groupA), groupB=_np(groupB))


def gen_topk(models, tu):
    """``eval_gc_allrec_onepep`` of the reference (training_utils.py:174-228) on the tiny model of
    the loop traces: for every peptide index k the list it returns (one rank: 1 + the number of
    receptors of the first aggregated batch that score higher than receptor k)."""
    import json
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
    from protein_clip_amd import synthetic as S
    from torch.utils.data import DataLoader
    tok = S.make_tokenizer()
    data = S.PairDataset(8, pep_len=(5, 12), rec_len=(10, 30), seed=7)
    torch.manual_seed(42)
    esm = S.make_esm(32, 2, 4)
    model = models.ExtendedCLIP(32, 16, 2, 2, 0.0, esm)
    loader = DataLoader(data, batch_size=4, shuffle=False, drop_last=True)
    out = {"batch_size": 4, "agg_batches": 2, "seed": 42, "ranks": {}, "diag": {}}
    for k in range(8):
        out["ranks"][str(k)] = [int(v) for v in tu.eval_gc_allrec_onepep(model, loader, "cpu", tok, agg_batches=2, k=k)]
    # the scores behind the ranks (peptide k against the eight receptors), to tell a near-tie from an error
    with torch.no_grad():
        rec = tok(data.rec, return_tensors="pt", padding=True)
        rec["temperature"] = model.temperature
        y = model.rec_encoder(rec)
        for k in range(8):
            pep = tok([data.pep[k]], return_tensors="pt", padding=True)
            pep["temperature"] = model.temperature
            x = model.pep_encoder(pep)
            out["diag"][str(k)] = [float(v) for v in torch.exp(x @ y.t())[0]]
    with open(os.path.join(OUT, "topk_trace.json"), "w") as f:
        json.dump(out, f, indent=1)


def gen_train_gc(models, tu):
    """The reference's own ``train_gc`` (training_utils.py:92-155) run through the replaying stand-in for
    ``grad_cache.functional.cached`` above, on the tiny model of the loop traces: 20 pairs in sub-batches of 4
    with ``accumulated_batches=2`` — two big batches of 8 and one trailing sub-batch that is encoded and dropped
    (:134) — for two epochs.  Stored: the loss of every big batch (:137), the epoch means the function returns
    (:155), and temperature and head parameters after each epoch, which pin the order of
    ``scaler.step / update / zero_grad`` (:151-153)."""
    import json
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
    from protein_clip_amd import synthetic as S
    from torch.utils.data import DataLoader
    tok = S.make_tokenizer()
    data = S.PairDataset(20, pep_len=(5, 12), rec_len=(10, 30), seed=11)
    torch.manual_seed(42)
    esm = S.make_esm(32, 2, 4)
    for prm in esm.parameters():
        prm.requires_grad = False
    model = models.ExtendedCLIP(32, 16, 2, 2, 0.0, esm)
    loader = DataLoader(data, batch_size=4, shuffle=False, drop_last=True)
    opt = torch.optim.Adam(model.parameters(), lr=1e-3)
    scaler = torch.cuda.amp.GradScaler(enabled=False)
    seen = []
    inner = tu._contrastive_loss_gc

    def recording_loss(x, y):
        loss = inner(x, y)
        seen.append(float(loss.item()))
        return loss
    tu._contrastive_loss_gc = recording_loss
    out = {"peptides": data.pep, "receptors": data.rec, "batch_size": 4, "accumulated_batches": 2, "seed": 42, "lr": 1e-3,
           "esm": {"hidden": 32, "layers": 2, "heads": 4},
           "head": {"input_dim": 32, "embedding_dim": 16, "h1": 2, "h2": 2, "dropout": 0.0}, "epochs": []}
    arrays = {}
    try:
        for ep in range(2):
            seen.clear()
            mean = tu.train_gc(model, loader, tok, opt, scaler, "cpu", accumulated_batches=2)
            out["epochs"].append({"mean": float(mean), "big_batch_losses": list(seen), "temperature": float(model.temperature.item())})
            for name, prm in model.named_parameters():
                if prm.requires_grad:
                    arrays[f"epoch{ep}/{name}"] = _np(prm).copy()   # a snapshot: the optimizer updates in place
    finally:
        tu._contrastive_loss_gc = inner
    with open(os.path.join(OUT, "train_gc_trace.json"), "w") as f:
        json.dump(out, f, indent=1)
    np.savez_compressed(os.path.join(OUT, "train_gc_params.npz"), **arrays)


def gen_head(models):
    """The reference's ``Encoder`` (models.py:7-90) with a stand-in trunk that returns fixed hidden
    states: everything after the per-residue network — masked mean, ``embedding_ffn``, normalise and
    temperature scale — with its gradients.  In train mode the dropout masks that ``nn.Dropout``
    drew inside ``embedding_ffn`` are recorded (output != 0) so the same elements can be dropped."""
    class Trunk(torch.nn.Module):
        def __init__(self, hidden):
            super().__init__()
            self.hidden = hidden

        def forward(self, input_ids=None, attention_mask=None):
            return types.SimpleNamespace(last_hidden_state=self.hidden)

    flat = {}
    for seed, (B, T, din, d), h2, p, train in [
        (0, (37, 19, 24, 128), 2, 0.1, True),     # the reference's head (h2 = 2, dropout 0.1), ragged batch
        (1, (16, 9, 16, 64), 3, 0.25, True),      # two blocks, narrower head
        (2, (8, 5, 16, 128), 1, 0.1, False),      # a single Linear, eval mode
        (3, (64, 12, 24, 128), 2, 0.1, False),    # eval mode: dropout is the identity
    ]:
        g = torch.Generator().manual_seed(seed)
        hidden = torch.randn(B, T, din, generator=g)
        lens = torch.randint(1, T + 1, (B,), generator=g)
        mask = (torch.arange(T)[None] < lens[:, None]).long()
        if seed == 0:
            mask[3] = 0                                # a sequence with no valid token: the count clamps to 1
        torch.manual_seed(100 + seed)
        enc = models.Encoder(din, d, 2, h2, p, Trunk(hidden))
        with torch.no_grad():                          # LayerNorm affine away from its (1, 0) initialisation
            for m in enc.embedding_ffn.modules():
                if isinstance(m, torch.nn.LayerNorm):
                    m.weight.add_(0.2 * torch.randn(d, generator=g)); m.bias.add_(0.1 * torch.randn(d, generator=g))
        enc.train(train)
        seen = {}

        def see_tokens(mod, inp, out):
            out.retain_grad()
            seen["tokens"] = out

        def see_pooled(mod, inp):
            inp[0].retain_grad()
            seen["pooled"] = inp[0]
        enc.amino_acid_ffn.register_forward_hook(see_tokens)
        enc.embedding_ffn.register_forward_pre_hook(see_pooled)
        keeps = []
        for m in enc.embedding_ffn.modules():
            if isinstance(m, torch.nn.Dropout):
                def hook(mod, inp, out):
                    assert not (inp[0] == 0).any()
                    keeps.append((out != 0).to(torch.uint8) if mod.training else torch.ones_like(out, dtype=torch.uint8))
                m.register_forward_hook(hook)
        t = torch.tensor(0.6 + 0.2 * seed, requires_grad=True)
        out = enc({"input_ids": None, "attention_mask": mask, "temperature": t})
        w = torch.randn(B, d, generator=g)
        (out * w).sum().backward()
        key = f"s{seed}_B{B}_T{T}_d{d}_h{h2}_{'train' if train else 'eval'}"
        c = dict(tokens=_np(seen["tokens"]), mask=_np(mask).astype(np.uint8), pooled=_np(seen["pooled"]), t=np.float32(t.item()),
                 p=np.float32(p), train=np.uint8(train), out=_np(out), d_out=_np(w), d_tokens=_np(seen["tokens"].grad),
                 d_pooled=_np(seen["pooled"].grad), d_t=_np(t.grad))
        if keeps:
            c["keep"] = _np(torch.stack(keeps))
        for name, prm in enc.embedding_ffn.named_parameters():
            c["param:" + name] = _np(prm)
            c["grad:" + name] = _np(prm.grad)
        flat.update({f"{key}/{f}": v for f, v in c.items()})
    np.savez_compressed(os.path.join(OUT, "head_tail.npz"), **flat)


def main():
    os.makedirs(OUT, exist_ok=True)
    sys.dont_write_bytecode = True
    _install_grad_cache_stand_in()
    sys.path.insert(0, REF)
    torch.manual_seed(0)
    torch.set_num_threads(8)
    import modules.models as models
    import modules.training_utils as tu
    import modules.training_utils_filip as tuf
    if "--head-only" in sys.argv:
        gen_head(models)
        print("head fixtures written", flush=True)
        return
    if "--b256-bf16n-only" in sys.argv:
        gen_filip_b256_bf16n(models, tuf)
        print("bf16n benchmark-size fixture written", flush=True)
        return
    if "--train-gc-only" in sys.argv:
        gen_train_gc(models, tu)
        print("train_gc trace written", flush=True)
        return
    if "--ties-topk-only" in sys.argv:
        gen_ties(models, tuf)
        gen_topk(models, tu)
        print("tie and top-k fixtures written", flush=True)
        return
    if "--loops-only" not in sys.argv:
        gen_ties(models, tuf)
        gen_topk(models, tu)
        gen_train_gc(models, tu)
        gen_head(models)
        gen_clip(tu)
        print("clip goldens written", flush=True)
        gen_filip(models, tuf)
        gen_filip_b256_bf16n(models, tuf)
        print("filip goldens written", flush=True)
    gen_loops(models, tu, tuf)
    print("loop traces written", flush=True)


if __name__ == "__main__":
    main()
