"""The C-ABI shared library loads without a GPU and exports every symbol include/rechub_hip.h declares.
No compute is launched here (argument validation only)."""
import ctypes
import os
import re

import pytest

from conftest import ROOT


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "rechub_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(rh_[a-z0-9_]+)\s*\(", text)))


def test_header_declares_the_hot_path():
    syms = declared_symbols()
    for must in ("rh_embed_fwd", "rh_embed_bwd", "rh_embed_scatter_rows", "rh_cross_fwd", "rh_cross_bwd",
                 "rh_adam_dense", "rh_seq_pool_fwd", "rh_fm_fwd", "rh_batch_gather"):
        assert must in syms


def test_library_exports_every_declared_symbol():
    from torch_rechub_amd import _lib
    lib = _lib.load()
    for name in declared_symbols():
        assert hasattr(lib, name), f"{name} declared in rechub_hip.h but not exported"
        assert name in _lib.SIGNATURES, f"{name} has no ctypes signature in _lib.SIGNATURES"
    assert lib.rh_abi_version() == _lib.ABI_VERSION


def test_argument_errors_are_reported_not_crashed():
    from torch_rechub_amd import _lib
    null = ctypes.c_void_p(0)
    with pytest.raises(RuntimeError, match="null descriptor"):
        _lib.call("rh_embed_fwd", null, null, 1, 4, 2, 16, null, 0, 32, null, 32, null, null, null, null, null, 0, null,
                  null)
    fake = ctypes.c_void_p(4096)  # never dereferenced: validation fails first
    with pytest.raises(RuntimeError, match="embed_dim 6 unsupported"):
        _lib.call("rh_embed_fwd", fake, fake, 1, 4, 2, 6, null, 0, 12, fake, 12, null, null, null, null, null, 0, null,
                  null)
    with pytest.raises(RuntimeError, match="layers per call unsupported"):
        _lib.call("rh_cross_fwd", fake, 8, fake, 8, fake, fake, 4, 8, 9, fake, 8, null)
    assert _lib.call("rh_cross_max_layers", 429) == 4
    assert _lib.call("rh_cross_max_layers", 4096) == 0
    assert _lib.call("rh_embed_bwd_nchunks", 4096, 0) == 16
    assert _lib.call("rh_cross_bwd_nblocks", 4096) == 256


def test_missing_library_fails_loudly(monkeypatch):
    from torch_rechub_amd import _lib
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", "/nonexistent/librechub_hip.so")
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        _lib.load()
