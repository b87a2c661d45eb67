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
    bound = set(_lib.SIGNATURES) | set(_lib.VALUE_FUNCS) | set(_lib.VALUE_FUNCS_I64) | {"lego_last_error", "lego_abi_version"}
    assert declared == bound, (declared - bound, bound - declared)


def test_prototypes_in_the_header_match_the_binding(handle):
    """argument COUNT and KIND (pointer / int / float / 64-bit) of every entry point, header vs ctypes binding"""
    protos = _lib.declared_prototypes()
    assert set(protos) == set(_lib.declared_symbols())            # every declaration was parsed
    for name, argtypes in list(_lib.SIGNATURES.items()) + list(_lib.VALUE_FUNCS.items()) + list(_lib.VALUE_FUNCS_I64.items()):
        ret, kinds = protos[name]
        assert ret == ("int64_t" if name in _lib.VALUE_FUNCS_I64 else "int"), name
        assert len(kinds) == len(argtypes), (name, len(kinds), len(argtypes))
        for i, (h, b) in enumerate(zip(kinds, argtypes)):
            assert h is b, (name, i, h, b)
    assert protos["lego_last_error"][1] == [] and protos["lego_abi_version"] == ("int", [])


def test_product_library_exports_nothing_undeclared(handle):
    """no tuning / debug hook ships in the product .so (they live behind `make tune`)"""
    import subprocess
    out = subprocess.run(["nm", "-D", "--defined-only", _lib.LIB_PATH], capture_output=True, text=True).stdout
    exported = {l.split()[-1] for l in out.splitlines() if " T " in l and l.split()[-1].startswith("lego_")}
    assert exported == set(_lib.declared_symbols()), exported ^ set(_lib.declared_symbols())


def test_host_code_under_address_sanitizer(tmp_path):
    """the host side of the C ABI built with -fsanitize=address (device code uninstrumented; GPU ASan is
    unavailable on this pool).  The argument-validation and error paths run without a GPU."""
    import subprocess
    import sys
    csrc = _lib.CSRC
    res = subprocess.run(["make", "-C", csrc, "asan", "-j4"], capture_output=True, text=True)
    assert res.returncode == 0, res.stderr[-2000:]
    rt = subprocess.run(["/opt/rocm/lib/llvm/bin/clang", "-print-file-name=libclang_rt.asan-x86_64.so"], capture_output=True,
                        text=True).stdout.strip()
    code = (
        "import ctypes\n"
        "from legommenders_amd import _lib\n"
        "h = _lib.lib()\n"
        "assert h.lego_abi_version() == _lib.ABI_VERSION\n"
        "rc = h.lego_adam_step(None, None, None, None, ctypes.c_int64(4), 1e-3, 0.9, 0.999, 1e-8, 0, 1.0, 0, None)\n"
        "assert rc != 0 and b'1-based' in h.lego_last_error()\n"
        "rc = h.lego_plan_batch(None, None, None, 0, 5, 50, None, None, 30, None, None, None, None, None, None, None)\n"
        "assert rc != 0 and b'bad sizes' in h.lego_last_error()\n"
        "rc = h.lego_linear_fwd(None, 3, None, 4, None, None, 4, 8, None, 4, 4, 0, None, None, None, None, None)\n"
        "assert rc != 0 and b'multiple of 4' in h.lego_last_error()\n"
        "rc = h.lego_sample_negatives(None, None, None, None, 100, 4, 99, 10, 1, 0, 0, 1, None, None, None)\n"
        "assert rc != 0 and b'unsupported' in h.lego_last_error()\n"
        "print('asan-ok')\n")
    env = dict(os.environ, LEGO_HIP_LIB=os.path.join(csrc, "liblego_hip_asan.so"), LD_PRELOAD=rt,
               ASAN_OPTIONS="detect_leaks=0:abort_on_error=1:protect_shadow_gap=0")
    run = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=env,
                         cwd=os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    assert run.returncode == 0 and "asan-ok" in run.stdout, (run.stdout[-500:], run.stderr[-3000:])
    assert "AddressSanitizer" not in run.stderr


def test_every_declared_symbol_is_exported(handle):
    for name in _lib.declared_symbols():
        assert hasattr(handle, name), name


def test_abi_version(handle):
    import re
    assert handle.lego_abi_version() == _lib.ABI_VERSION
    assert int(re.search(r"#define LEGO_ABI_VERSION (\d+)", open(_lib.HEADER).read()).group(1)) == _lib.ABI_VERSION


def test_stale_library_is_refused(monkeypatch):
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "ABI_VERSION", _lib.ABI_VERSION + 1)
    with pytest.raises(_lib.LegoHipError, match="rebuild"):
        _lib.lib()


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", str(tmp_path / "nope.so"))
    with pytest.raises(_lib.LegoHipError):
        _lib.lib()


def test_argument_validation_without_gpu(handle):
    # argument checks run before any kernel launch, so they are testable on CPU
    rc = handle.lego_adam_step(None, None, None, None, ctypes.c_int64(4), 1e-3, 0.9, 0.999, 1e-8, 0, 1.0, 0, None)
    assert rc != 0 and b"1-based" in handle.lego_last_error()
