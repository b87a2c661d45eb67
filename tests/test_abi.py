"""CPU-side checks of the C-ABI boundary: the library builds/loads and exports every symbol
that include/lego_hip.h declares (no compute calls without a GPU)."""
import ctypes
import os

import pytest

from legommenders_amd import _lib


@pytest.fixture(scope="module")
def handle():
    if not os.path.exists(_lib.LIB_PATH):
        _lib.build()
    return _lib.lib()


def test_header_and_binding_agree(handle):
    declared = set(_lib.declared_symbols())
    bound = set(_lib.SIGNATURES) | {"lego_last_error", "lego_abi_version"}
    assert declared == bound, (declared - bound, bound - declared)


def test_every_declared_symbol_is_exported(handle):
    for name in _lib.declared_symbols():
        assert hasattr(handle, name), name


def test_abi_version(handle):
    assert handle.lego_abi_version() == 1


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", str(tmp_path / "nope.so"))
    with pytest.raises(_lib.LegoHipError):
        _lib.lib()


def test_argument_validation_without_gpu(handle):
    # argument checks run before any kernel launch, so they are testable on CPU
    rc = handle.lego_adam_step(None, None, None, None, ctypes.c_int64(4), 1e-3, 0.9, 0.999, 1e-8, 0, 1.0, 0, None)
    assert rc != 0 and b"1-based" in handle.lego_last_error()
