"""CPU tests of the host-side Python: no CPU fallback, shape checks, checkpoint key names and
module structure against what the reference's own models produce (golden key list)."""
import json
import os

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def trace():
    with open(os.path.join(ROOT, "tests", "golden", "loops_trace.json")) as f:
        return json.load(f)


def test_cpu_tensors_are_rejected_not_silently_computed():
    from protein_clip_amd import kernels as K, ops
    from protein_clip_amd._lib import PclipError
    with pytest.raises(PclipError):
        K.l2norm_scale_fwd(torch.randn(4, 8), torch.tensor([1.0]))
    with pytest.raises(PclipError):
        ops.clip_loss(torch.randn(4, 8), torch.randn(4, 8))
    with pytest.raises(PclipError):
        ops.filip_similarity(torch.randn(2, 5, 8), torch.randn(2, 6, 8), torch.ones(2, 5), torch.ones(2, 6), 1.0)
    with pytest.raises(PclipError):
        ops.row_ce(torch.randn(4, 4))


def test_operator_shape_checks():
    from protein_clip_amd import ops
    with pytest.raises(ValueError):
        ops.clip_loss(torch.randn(4, 8), torch.randn(5, 8))
    with pytest.raises(ValueError):
        ops.filip_similarity(torch.randn(2, 5, 8), torch.randn(2, 6, 4), torch.ones(2, 5), torch.ones(2, 6), 1.0)
    with pytest.raises(ValueError):
        ops.filip_similarity(torch.randn(2, 5, 8), torch.randn(2, 6, 8), torch.ones(2, 4), torch.ones(2, 6), 1.0)
    with pytest.raises(ValueError):
        ops.row_ce(torch.randn(4))


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    from protein_clip_amd import _lib
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", str(tmp_path / "absent.so"))
    with pytest.raises(_lib.PclipError, match="no CPU fallback"):
        _lib.load()


@pytest.mark.parametrize("kind", ["clip", "filip"])
def test_state_dict_keys_match_the_reference_models(trace, kind):
    from protein_clip_amd import synthetic as S
    from protein_clip_amd.modules import models
    cfg, head = trace["esm"], trace["head"]
    esm = S.make_esm(cfg["hidden"], cfg["layers"], cfg["heads"])
    cls = models.ExtendedCLIP if kind == "clip" else models.ExtendedFILIP
    model = cls(head["input_dim"], head["embedding_dim"], head["h1"], head["h2"], head["dropout"], esm)
    assert list(model.state_dict().keys()) == trace[kind]["state_dict_keys"]
    assert float(model.temperature) == 1.0


def test_same_seed_gives_the_reference_initialisation_order():
    """Sub-modules are created in the reference's order, so one seed initialises both
    implementations identically (what the loop-trace parity test on the GPU relies on)."""
    from protein_clip_amd.modules import models
    torch.manual_seed(3)
    a = models.Encoder(8, 4, 2, 3, 0.1, None)
    torch.manual_seed(3)
    proj = torch.nn.Linear(8, 4)
    assert torch.equal(a.projection.weight, proj.weight)
    assert [type(m).__name__ for m in a.amino_acid_ffn[0]] == ["Linear", "ReLU", "LayerNorm", "Dropout"]
    assert isinstance(a.amino_acid_ffn[1], torch.nn.Linear) and len(a.embedding_ffn) == 3


def test_list_arguments_are_concatenated():
    from protein_clip_amd.modules import training_utils as tu
    parts = [torch.ones(2, 3), torch.zeros(1, 3)]
    assert tu._cat(parts).shape == (3, 3)
    t = torch.ones(4, 3)
    assert tu._cat(t) is t
