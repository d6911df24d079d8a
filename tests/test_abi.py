"""The C-ABI library loads on a CPU-only box and exports every symbol include/neurofluid_hip.h declares
(no compute calls here)."""
import ctypes
import os
import re

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    txt = open(os.path.join(REPO, "include", "neurofluid_hip.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(nf_[a-z0-9_]+)\s*\(", txt)))


@pytest.fixture(scope="module")
def lib_path():
    from neurofluid_amd import build
    return build.build()      # hipcc cross-compiles gfx950 without a GPU


def test_header_symbols_exported(lib_path):
    lib = ctypes.CDLL(lib_path)
    names = declared_symbols()
    assert len(names) >= 25
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/neurofluid_hip.h but not exported"


def test_ctypes_prototypes_cover_header(lib_path):
    from neurofluid_amd import _lib
    assert set(declared_symbols()) == set(_lib.PROTOTYPES), set(declared_symbols()) ^ set(_lib.PROTOTYPES)
    lib = _lib.load()
    assert lib.nf_version() == 100
    assert lib.nf_last_error() is not None


def test_host_side_queries(lib_path):
    """Pure host entry points (no GPU needed): sizes and argument validation."""
    from neurofluid_amd import _lib
    lib = _lib.load()
    bb = (ctypes.c_float * 6)(-1, -1, -1, 1, 1, 2.5)
    n1 = lib.nf_grid_workspace_bytes(5000, 0.225, bb)
    n2 = lib.nf_grid_workspace_bytes(50000, 0.1125, bb)
    assert 0 < n1 < n2
    assert lib.nf_grid_workspace_bytes(10, -1.0, bb) == 0
    assert lib.nf_nerf_packed_floats(198, 54) > 668420          # padded weights + biases
    cx, cd, qx, qd = ctypes.c_int(), ctypes.c_int(), ctypes.c_int(), ctypes.c_int()
    lib.nf_render_feature_dims(15, ctypes.byref(cx), ctypes.byref(cd), ctypes.byref(qx), ctypes.byref(qd))
    assert (cx.value, cd.value, qx.value, qd.value) == (198, 54, 25, 7)
    lib.nf_render_feature_dims(0, ctypes.byref(cx), ctypes.byref(cd), ctypes.byref(qx), ctypes.byref(qd))
    assert (cx.value, cd.value) == (63, 27)
    # bad arguments come back as error codes with a message, never abort
    rc = lib.nf_ball_query_firstk(None, None, None, 4, 0.1, 20, None, None, None, None)
    assert rc < 0 and b"null" in lib.nf_last_error()


def test_product_has_no_oracle_import():
    """The product package must never import the oracle (parity claims depend on it)."""
    pkg = os.path.join(REPO, "neurofluid_amd")
    for root, _, files in os.walk(pkg):
        for f in files:
            if f.endswith(".py"):
                src = open(os.path.join(root, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", src, flags=re.M), f
