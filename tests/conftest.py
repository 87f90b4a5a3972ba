"""Shared test configuration: registers the ``gpu`` marker, puts the repository root on
``sys.path`` and offers loaders for the golden vectors under ``tests/golden/``."""
import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def is_port_clash(exc) -> bool:
    """True when a multi-process rendezvous failed because the port picked for it was taken in the meantime
    (EADDRINUSE, errno 98): the ONLY failure the two-rank fixtures try again on a fresh port; anything else a
    rank raises is a real failure and surfaces on the first attempt."""
    text = repr(exc) + str(exc)
    return any(m in text for m in ("Address already in use", "EADDRINUSE", "errno: 98", "Errno 98", "address already in use"))


def load_cases(name):
    """Return {case_key: {field: torch tensor}} from a grouped ``.npz`` (keys 'case/field')."""
    z = np.load(os.path.join(GOLDEN, name), allow_pickle=False)
    out = {}
    for k in z.files:
        case, field = k.split("/")
        out.setdefault(case, {})[field] = torch.from_numpy(np.asarray(z[k]))
    return out


def load_flat(name):
    z = np.load(os.path.join(GOLDEN, name), allow_pickle=False)
    return {k: torch.from_numpy(np.asarray(z[k])) for k in z.files}


@pytest.fixture(scope="session")
def clip_cases():
    return load_cases("clip_small.npz")


@pytest.fixture(scope="session")
def filip_cases():
    return load_cases("filip_small.npz")


def head_case_params(c):
    """The parameter list of a head fixture case in module order (W, b, gamma, beta per block, then the
    last W, b), the matching gradients, the dropout masks (or None) and 1 / (1 - p)."""
    names = sorted((k[6:] for k in c if k.startswith("param:")), key=lambda n: [int(v) if v.isdigit() else v for v in n.split(".")])
    # names: "0.0.weight", "0.0.bias", "0.2.weight", "0.2.bias", ..., "<last>.weight", "<last>.bias"
    order = sorted(names, key=lambda n: (int(n.split(".")[0]), n.split(".")[1] if n.count(".") == 2 else "", 0 if n.endswith("weight") else 1))
    params = [c["param:" + n] for n in order]
    grads = [c["grad:" + n] for n in order]
    train = bool(c["train"].item())
    keep = c["keep"] if (train and "keep" in c) else None
    scale = 1.0 / (1.0 - float(c["p"])) if train else 1.0
    return params, grads, keep, scale


@pytest.fixture(scope="session")
def head_cases():
    return load_cases("head_tail.npz")
