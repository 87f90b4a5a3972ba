"""CPU checks of the C-ABI boundary: every function declared in include/pclip.h is exported by
the built library and bound (with argument types) by the ctypes layer, and argument
validation rejects bad calls before anything is launched.  No GPU needed."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    text = open(os.path.join(ROOT, "include", "pclip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(pclip_[a-z0-9_]+)\s*\(", text)))


@pytest.fixture(scope="module")
def lib():
    from protein_clip_amd import _lib
    return _lib.load()


def test_header_library_and_binding_agree(lib):
    from protein_clip_amd import _lib
    declared = _declared()
    assert len(declared) >= 17
    assert sorted(_lib.EXPORTED_SYMBOLS) == declared
    raw = ctypes.CDLL(_lib.LIB_PATH)
    for name in declared:
        assert hasattr(raw, name), f"{name} is declared in pclip.h but not exported by libpclip.so"


def test_version_and_workspace_queries(lib):
    assert lib.pclip_version() >= 1
    assert lib.pclip_clip_workspace_bytes(256, 256, 128) > 0
    assert lib.pclip_clip_workspace_bytes(1024, 8192, 128) >= 2 * 1024 * 128 * 4
    assert lib.pclip_clip_step_workspace_bytes(256, 128) > lib.pclip_clip_workspace_bytes(256, 256, 128)
    assert lib.pclip_clip_workspace_bytes(0, 0, 0) == 0
    # bucket lists: one u32 per (pair, token) on each side
    assert lib.pclip_filip_bwd_workspace_bytes(256, 256, 256, 256) >= 2 * 256 * 256 * 256 * 4


def test_bad_arguments_are_rejected_before_launch(lib):
    assert lib.pclip_l2norm_scale_fwd(None, None, None, 4, 8, None) == -1
    assert b"null pointer" in lib.pclip_last_error()
    one = ctypes.c_void_p(16)   # never dereferenced: the size checks fire first
    assert lib.pclip_l2norm_scale_fwd(one, one, one, 0, 8, None) == -1
    assert lib.pclip_clip_fwd(one, one, one, one, 8, 8, 130, 0, one, one, one, None, 0, one, 1 << 20, None) == -1
    assert b"d=130" in lib.pclip_last_error()
    assert lib.pclip_clip_fwd(one, one, one, one, 8, 4, 128, 0, one, one, one, None, 0, one, 1 << 20, None) == -1   # Br > Bg
    assert lib.pclip_clip_fwd(one, one, one, one, 8, 8, 128, 0, one, one, one, None, 0, one, 16, None) == -1        # workspace too small
    assert b"workspace" in lib.pclip_last_error()
    assert lib.pclip_filip_prep(one, one, 4, 5000, 128, 1, None, one, one, None) == -1                                # T > 4096
    assert lib.pclip_filip_prep(one, one, 4, 64, 129, 1, None, one, one, None) == -1                                  # d > 128
    assert lib.pclip_clip_step_noalloc(one, one, one, 4096, 128, one, one, one, one, None) == -1                      # beyond the built-in scratch
    assert b"built-in buffer" in lib.pclip_last_error()
    assert lib.pclip_rowce_fwd(one, 8, 4, 0, 8, 0, one, one, None) == -1                                              # target column out of range
    # FILIP backward: the gather addresses partner rows with 32-bit byte offsets, so one side's images stay below 4 GB
    args = [one, one, 60000, 512, one, one, 8, 64, one, one, one, one, one, one, one, one, one, one, one, 1 << 20, None]
    assert lib.pclip_filip_sim_bwd(*args) == -1
    assert b"exceed 4 GB" in lib.pclip_last_error()
