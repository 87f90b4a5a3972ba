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
