"""GPU tests of the drop-in Python surface (modules.models / training_utils /
training_utils_filip) on top of the HIP operators: the reference's own two-epoch loop traces,
autograd through the operators, and the big-batch (GradCache-semantics) trainer."""
import json
import os

import pytest
import torch
from torch.utils.data import DataLoader

from conftest import head_case_params
from oracle import pclip_oracle as O

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def env():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from protein_clip_amd import synthetic as S
    from protein_clip_amd.modules import models, training_utils, training_utils_filip
    with open(os.path.join(ROOT, "tests", "golden", "loops_trace.json")) as f:
        trace = json.load(f)
    return dict(S=S, models=models, tu=training_utils, tuf=training_utils_filip, trace=trace, tok=S.make_tokenizer())


def _build(env, kind, dropout=None):
    tr = env["trace"]
    torch.manual_seed(tr["seed"])
    esm = env["S"].make_esm(tr["esm"]["hidden"], tr["esm"]["layers"], tr["esm"]["heads"])
    for prm in esm.parameters():
        prm.requires_grad = False
    h = tr["head"]
    cls = env["models"].ExtendedCLIP if kind == "clip" else env["models"].ExtendedFILIP
    return cls(h["input_dim"], h["embedding_dim"], h["h1"], h["h2"], h["dropout"] if dropout is None else dropout, esm).cuda()


class _Pairs(torch.utils.data.Dataset):
    def __init__(self, pep, rec):
        self.pep, self.rec = pep, rec

    def __len__(self):
        return len(self.pep)

    def __getitem__(self, i):
        return self.pep[i], self.rec[i]


@pytest.mark.parametrize("kind,tol", [("clip", 2e-4), ("filip", 3e-3)])
def test_two_epoch_trace_matches_the_reference_loops(env, kind, tol):
    """Same seed, same synthetic pairs, same Adam: per-epoch train / eval means and the learnt
    temperature as produced by the reference's train()/evaluate() on CPU."""
    tr = env["trace"]
    model = _build(env, kind)
    loader = DataLoader(_Pairs(tr["peptides"], tr["receptors"]), batch_size=tr["batch_size"], shuffle=False, drop_last=True)
    opt = torch.optim.Adam(model.parameters(), lr=tr["lr"])
    mod = env["tu"] if kind == "clip" else env["tuf"]
    for want in tr[kind]["epochs"]:
        got_tr = mod.train(model, loader, opt, env["tok"], "cuda")
        got_ev = mod.evaluate(model, loader, env["tok"], "cuda")
        assert abs(got_tr - want["train"]) < tol, (got_tr, want)
        assert abs(got_ev - want["eval"]) < tol, (got_ev, want)
        assert abs(model.temperature.item() - want["temperature"]) < tol


def test_clip_operators_autograd(env, clip_cases):
    from protein_clip_amd import ops
    c = clip_cases["s2_B32_d128"]
    e_pep = c["e_pep"].cuda().requires_grad_(True)
    e_rec = c["e_rec"].cuda().requires_grad_(True)
    t = c["t"].cuda().requires_grad_(True)
    x, y = ops.l2norm_scale(e_pep, t), ops.l2norm_scale(e_rec, t)
    loss = env["tu"]._contrastive_loss(x, y.t())
    loss.backward()
    torch.testing.assert_close(loss.cpu(), c["loss"], rtol=1e-5, atol=1e-6)
    torch.testing.assert_close(e_pep.grad.cpu(), c["d_e_pep"], rtol=1e-4, atol=2e-7)
    torch.testing.assert_close(e_rec.grad.cpu(), c["d_e_rec"], rtol=1e-4, atol=2e-7)
    torch.testing.assert_close(t.grad.cpu(), c["d_t"], rtol=1e-4, atol=2e-6)
    xs, ys = list(torch.chunk(x.detach(), 4)), list(torch.chunk(y.detach(), 4))
    torch.testing.assert_close(env["tu"]._contrastive_loss_gc(xs, ys).cpu(), c["loss_gc4"], rtol=1e-5, atol=1e-6)
    torch.testing.assert_close(env["tu"].get_logits(xs, ys).cpu(), c["exp_logits"], rtol=1e-5, atol=1e-6)


def test_filip_score_function_and_loss_autograd(env, filip_cases):
    FIL = env["models"].ExtendedFILIP
    for key in ("s3_B8_TA48_TB40_d128_bf16n", "s4_B5_TA70_TB33_d128_bf16n"):
        c = filip_cases[key]
        a = c["hA"].cuda().requires_grad_(True)
        b = c["hB"].cuda().requires_grad_(True)
        t = c["t"].cuda().requires_grad_(True)
        sA, sB = FIL._filip_similarity_score(a, b, c["maskA"].cuda(), c["maskB"].cuda(), t)
        loss = env["tuf"]._compute_loss(sA, sB)
        loss.backward()
        torch.testing.assert_close(sA.detach().cpu(), c["sim_A"], rtol=2e-6, atol=2e-6)
        torch.testing.assert_close(sB.detach().cpu(), c["sim_B"], rtol=2e-6, atol=2e-6)
        torch.testing.assert_close(loss.detach().cpu(), c["loss"], rtol=1e-5, atol=1e-6)
        torch.testing.assert_close(t.grad.cpu(), c["d_t"], rtol=1e-4, atol=1e-6)
        for got, want in ((a.grad.cpu(), c["d_hA"]), (b.grad.cpu(), c["d_hB"])):
            bad = ((got - want).abs() > 1e-6 + 1e-3 * want.abs()).float().mean().item()
            assert bad == 0.0   # measured: none outside, relative L2 error 1e-7
            assert ((got - want).norm() / want.norm()).item() < 1e-5
        torch.testing.assert_close(env["tuf"]._contrastive_loss(c["sim_B"].cuda(), use_dcl=True).cpu(), c["loss_B_dcl"], rtol=1e-5, atol=1e-6)


def test_filip_group_axis_variant(env, filip_cases):
    c = filip_cases["s1_B6_TA9_TB11_d16_raw"]
    FIL = env["models"].ExtendedFILIP
    hA = O.bf16_round(O.filip_normalise(c["hA"])).cuda()
    hB = O.bf16_round(O.filip_normalise(c["hB"])).cuda()
    mA, mB = c["maskA"].cuda(), c["maskB"].cuda()
    flat = FIL._filip_similarity_score(hA, hB, mA, mB, 1.0)
    grouped = FIL._filip_similarity_score(hA.reshape(2, 3, 9, 16), hB.reshape(3, 2, 11, 16), mA.reshape(2, 3, 9), mB.reshape(3, 2, 11), 1.0,
                                          include_group=True)
    assert grouped[0].shape == (2, 3, 3, 2)
    torch.testing.assert_close(grouped[0][1, 2], flat[0][3:6, 4:6])
    torch.testing.assert_close(grouped[1][0, 1], flat[1][0:3, 2:4])


def test_topk_eval_matches_the_reference_ranks(env):
    """``eval_gc_allrec_onepep`` (training_utils.py:174-228 of the reference) on the exp-logits
    kernel: the rank list the reference returns for every peptide index k of the tiny model."""
    with open(os.path.join(ROOT, "tests", "golden", "topk_trace.json")) as f:
        want = json.load(f)
    S = env["S"]
    data = S.PairDataset(8, pep_len=(5, 12), rec_len=(10, 30), seed=7)
    torch.manual_seed(want["seed"])
    model = env["models"].ExtendedCLIP(32, 16, 2, 2, 0.0, S.make_esm(32, 2, 4)).cuda()
    loader = DataLoader(data, batch_size=want["batch_size"], shuffle=False, drop_last=True)
    for k in range(8):
        got = env["tu"].eval_gc_allrec_onepep(model, loader, "cuda", env["tok"], agg_batches=want["agg_batches"], k=k)
        assert got == want["ranks"][str(k)], (k, got, want["ranks"][str(k)], want["diag"][str(k)])


def test_big_batch_trainer_equals_one_big_batch(env):
    """train_gc with 2 sub-batches of 4 must take the same optimizer step as one plain step on
    the 8 pairs at once (dropout off): same loss, same updated head weights."""
    tr = env["trace"]
    data = _Pairs(tr["peptides"], tr["receptors"])
    m1, m2 = _build(env, "clip"), _build(env, "clip")
    m2.load_state_dict(m1.state_dict())
    o1 = torch.optim.SGD([p for p in m1.parameters() if p.requires_grad], lr=0.1)
    o2 = torch.optim.SGD([p for p in m2.parameters() if p.requires_grad], lr=0.1)
    scaler = torch.amp.GradScaler("cuda", enabled=False)
    loss_gc = env["tu"].train_gc(m1, DataLoader(data, batch_size=4, shuffle=False), env["tok"], o1, scaler, "cuda", accumulated_batches=2)
    loss_big = env["tu"].train(m2, DataLoader(data, batch_size=8, shuffle=False), o2, env["tok"], "cuda")
    assert abs(loss_gc - loss_big) < 2e-5
    for (n1, p1), (n2, p2) in zip(m1.named_parameters(), m2.named_parameters()):
        if p1.requires_grad:
            torch.testing.assert_close(p1, p2, rtol=2e-4, atol=2e-6, msg=lambda m: f"{n1}: {m}")


def test_big_batch_trainer_matches_the_reference_train_gc(env):
    """``train_gc`` on the HIP operators against what the REFERENCE's own ``train_gc`` (training_utils.py:92-155, run
    through a replaying stand-in for ``grad_cache.functional.cached`` by oracle/gen_golden.py) returned and left
    behind: every big-batch loss, the epoch means, temperature and all head parameters after each of two epochs;
    five sub-batches with two per big batch, so the trailing one is encoded and dropped as in the reference."""
    from test_oracle_golden import run_train_gc_against_trace
    run_train_gc_against_trace("cuda", 2e-4, 2e-2)


def test_big_batch_trainer_replays_dropout(env):
    """With dropout on, the second (graph) pass of a sub-batch must see the first pass's masks:
    the representation recomputed in the replay equals the cached one."""
    model = _build(env, "clip", dropout=0.5)
    model.train()
    tok = env["tok"]
    batch = tok(env["trace"]["peptides"][:4], return_tensors="pt", padding=True).to("cuda")
    batch["temperature"] = model.temperature
    rep, snap = env["tu"]._encode_cached(model.pep_encoder, batch, "cuda")
    keep = env["tu"]._RngSnapshot("cuda")
    snap.restore()
    again = model.pep_encoder(batch)
    keep.restore()
    torch.testing.assert_close(again.detach(), rep.detach(), rtol=0, atol=0)
    other = model.pep_encoder(batch)   # a fresh pass draws new masks
    assert not torch.equal(other.detach(), rep.detach())


@pytest.mark.parametrize("script,extra", [("main", ["--accumulated-batches", "2"]), ("main_2protein_filip", ["--batch-size", "8"])])
def test_entry_scripts_run_offline(env, script, extra, monkeypatch, tmp_path):
    """The two entry scripts run end to end without the network (random two-layer trunk, random
    pairs): one epoch, finite losses, the reference's CSV header and a checkpoint whose keys
    are the reference's state-dict keys."""
    import importlib
    monkeypatch.chdir(tmp_path)
    monkeypatch.setattr("sys.argv", [script + ".py", "--synthetic", "--epochs", "1", "--train-pairs", "32", "--eval-pairs", "16",
                                     "--trunk-layers", "2", "--max-len", "60"] + extra)
    mod = importlib.import_module(script)
    train_losses, val_losses, test_loss = mod.main()
    assert len(train_losses) == 1 and all(map(lambda v: v == v and abs(v) < 1e3, train_losses + val_losses + [test_loss]))
    run_dir = next((tmp_path / "runs").iterdir())
    assert (run_dir / "losses_per_epoch.txt").read_text().startswith("Epoch,Train Loss,Validation Loss\n1,")
    keys = set(torch.load(run_dir / "best_model.pth", weights_only=True).keys())
    kind = "clip" if script == "main" else "filip"
    head_keys = {k for k in env["trace"][kind]["state_dict_keys"] if "esm_model" not in k}
    assert head_keys and head_keys <= keys


def test_operators_accept_half_precision_activations(env):
    """A caller under autocast (the reference's train_gc decorates its loss with @autocast) hands
    the operators fp16 / bf16 activations: they are upcast through autograd and the gradients come
    back in the caller's dtype."""
    from protein_clip_amd import ops
    g = torch.Generator().manual_seed(3)
    t = torch.tensor(1.0, device="cuda", requires_grad=True)
    for dt in (torch.float16, torch.bfloat16):
        e1 = torch.randn(16, 128, generator=g).cuda().to(dt).requires_grad_(True)
        e2 = torch.randn(16, 128, generator=g).cuda().to(dt).requires_grad_(True)
        loss = ops.clip_loss(ops.l2norm_scale(e1, t), ops.l2norm_scale(e2, t))
        hA = torch.randn(4, 20, 128, generator=g).cuda().to(dt).requires_grad_(True)
        hB = torch.randn(4, 24, 128, generator=g).cuda().to(dt).requires_grad_(True)
        mA, mB = torch.ones(4, 20, device="cuda"), torch.ones(4, 24, device="cuda")
        sA, sB = ops.filip_similarity(hA, hB, mA, mB, t, normalise=True)
        total = loss + (ops.row_ce(sA) + ops.row_ce(sB)) / 2 + ops.masked_mean(hA, mA).sum()
        total.backward()
        assert total.dtype == torch.float32 and torch.isfinite(total)
        for x in (e1, e2, hA, hB):
            assert x.grad is not None and x.grad.dtype == dt and torch.isfinite(x.grad.float()).all()


def test_encoder_tail_matches_the_reference_encoder(env, head_cases, monkeypatch):
    """``Encoder.forward`` from the per-residue features on (models.py:48-51) against the reference's
    Encoder run with a stand-in trunk: the fused head is what ``forward`` calls, in eval mode and in
    train mode with the dropout masks the reference drew (fed through the ``torch.rand`` the tower
    draws its masks from)."""
    import types
    models = env["models"]
    for key, c in head_cases.items():
        params, grads, keep, scale = head_case_params(c)
        d, h2 = c["pooled"].shape[1], (len(params) - 2) // 4 + 1
        tokens = c["tokens"].cuda().requires_grad_(True)

        class Trunk(torch.nn.Module):
            def forward(self, input_ids=None, attention_mask=None):
                return types.SimpleNamespace(last_hidden_state=tokens)
        enc = models.Encoder(d, d, 1, h2, float(c["p"]), Trunk()).cuda()
        enc.projection = torch.nn.Identity()
        enc.amino_acid_ffn = torch.nn.Identity()
        assert models._head_layout(enc.embedding_ffn) is not None
        with torch.no_grad():
            for prm, val in zip(models._head_layout(enc.embedding_ffn)[0], params):
                prm.copy_(val)
        enc.train(bool(c["train"].item()))
        if keep is not None:
            monkeypatch.setattr(torch, "rand", lambda shape, device=None: keep.to(device=device, dtype=torch.float32))
        t = c["t"].clone().cuda().requires_grad_(True)
        out = enc({"input_ids": None, "attention_mask": c["mask"].cuda(), "temperature": t})
        monkeypatch.undo()
        torch.testing.assert_close(out.cpu(), c["out"], rtol=1e-4, atol=1e-6)
        (out * c["d_out"].cuda()).sum().backward()
        torch.testing.assert_close(tokens.grad.cpu(), c["d_tokens"], rtol=2e-4, atol=2e-6)
        torch.testing.assert_close(t.grad.cpu(), c["d_t"], rtol=2e-4, atol=1e-5)
        for prm, want in zip(models._head_layout(enc.embedding_ffn)[0], grads):
            torch.testing.assert_close(prm.grad.cpu(), want, rtol=2e-4, atol=2e-5)
