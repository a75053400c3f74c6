"""CPU: the C-ABI library builds for gfx950, loads, and exports every symbol that
include/mal_hip.h declares; argument validation returns error codes (no kernel is launched)."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    from mal_amd import build, _lib
    build.build(verbose=False)
    return _lib.load()


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "mal_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(mal_[a-z0-9_]+)\s*\(", text)))


def test_every_declared_symbol_is_exported(lib):
    from mal_amd import _lib
    names = declared_symbols()
    assert len(names) >= 25
    raw = ctypes.CDLL(_lib.LIB_PATH)
    for n in names:
        assert hasattr(raw, n), "libmal_hip.so does not export %s" % n
        assert n in _lib.SIGNATURES, "mal_amd/_lib.py has no ctypes signature for %s" % n
    assert sorted(_lib.SIGNATURES) == names


def test_version_strerror_workspace(lib):
    assert lib.mal_version() == 100
    assert b"workspace" in lib.mal_strerror(-3)
    assert lib.mal_strerror(0) == b"ok"
    assert lib.mal_workspace_bytes(12, 192, 640) > 0
    assert lib.mal_workspace_bytes(0, 192, 640) == 0
    # grows with the tile count
    assert lib.mal_workspace_bytes(12, 192, 640) < lib.mal_workspace_bytes(24, 192, 640)


def test_argument_validation_without_device(lib):
    n = None
    # null pointers / bad shapes are rejected before any HIP call
    assert lib.mal_disp_to_depth(n, 16, 0.1, 100.0, n, n, n) == -1
    assert lib.mal_backproject(n, n, 1, 4, 4, n, n) == -1
    assert lib.mal_backproject(n, n, 1, 1, 4, n, n) == -2        # reflection pad needs >= 2
    assert lib.mal_ssim(n, n, 1, 3, 1, 8, n, n) == -2
    assert lib.mal_smooth_loss(n, n, 1, 3, 8, 8, 1, n, n, n, 0, n) == -1
    assert lib.mal_axpy_maps(0, n, n, n, n, n, 10, n, 0, n) == -1
    assert lib.mal_sum_f64(n, 0, n, n, 0, n) == -1


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    from mal_amd import _lib
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", str(tmp_path / "nope.so"))
    with pytest.raises(_lib.MalError):
        _lib.load()


def test_cpu_tensors_are_rejected(lib):
    import torch
    from mal_amd import _lib, layers
    with pytest.raises(_lib.MalError):
        layers.SSIM()(torch.rand(1, 3, 8, 8), torch.rand(1, 3, 8, 8))
