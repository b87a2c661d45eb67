"""ctypes binding of liblego_hip.so (the C ABI declared in include/lego_hip.h).

The product path has NO CPU fallback: if the shared object is missing or does not export a
declared symbol, importing / calling raises.  `build()` compiles it in-tree with hipcc for gfx950.
"""
from __future__ import annotations

import ctypes
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(_HERE, "csrc")
LIB_PATH = os.environ.get("LEGO_HIP_LIB") or os.path.join(CSRC, "liblego_hip.so")   # override: A/B of two builds in one run
HEADER = os.path.join(os.path.dirname(_HERE), "include", "lego_hip.h")

P = ctypes.c_void_p
I = ctypes.c_int
F = ctypes.c_float
I64 = ctypes.c_int64
U64 = ctypes.c_uint64
U32 = ctypes.c_uint32


ABI_VERSION = 8      # == LEGO_ABI_VERSION of include/lego_hip.h this binding was written against


class LegoDropout(ctypes.Structure):
    _fields_ = [("p", ctypes.c_float), ("seed", ctypes.c_uint64), ("site", ctypes.c_uint32), ("mask", ctypes.c_void_p)]


# name -> argtypes (all return int; 0 = ok).  Mirrors include/lego_hip.h one to one.
SIGNATURES = {
    "lego_plan_batch": [P, P, P, I, I, I, P, P, I, P, P, P, P, P, P, P],
    "lego_plan_dense": [P, I, I, P, P, P, P],
    "lego_gather_rows": [P, I, I, P, I, P, P, I, I, P],
    "lego_nrms_decode_rows": [P, I, P, P, P, P, P, P],
    "lego_nrms_key_rows": [P, I, P, I, P, P],
    "lego_nrms_decode_keys": [P, I, P, I, P, P, P, P, P],
    "lego_nrms_special_grads": [P, I, P, P, P, I, I, P, P, I, I, P],
    "lego_mask_dropout_rows": [P, I, I, P, I, P, P, P, P],
    "lego_nrms_user_head_train": [P, I, P, P, P, I, I, I, I, F, P, I, P, P, P, I, P, I, P, I, P, P],
    "lego_attn_fold_prepare": [P, P, P, P, P, P, P, P, P, P, I, I, P],
    "lego_attn_fold_grads": [P, P, P, P, P, P, P, P, P, P, P, P, P, P, P, P, I, I, P],
    "lego_scatter_add_rows": [P, I, I, I, P, I, P, P, I, P],
    "lego_unique_tokens": [P, I, P, I, P, U32, P, P, P, P, P, P, P, P, P, P],
    "lego_sort_rows": [P, I, P, P, P, I64, P],
    "lego_expand_rows": [P, I, P, I, P, I, P, P, P, I, P, P, I, P, P, I, P],
    "lego_qkv_expand_dropcorr": [P, I, P, I, P, I, P, P, P, P, I, P, I, I, P, I, P],
    "lego_segment_sum_rows": [P, I, I, P, P, I, P, P, P, I, I, P, I, P, P, P],
    "lego_zero_rows": [P, I, I, I, P, P],
    "lego_scatter_add_rows_range": [P, I, I, P, I, P, P, I, I, I, P],
    "lego_linear_fwd": [P, I, P, I, P, P, I, I, P, I, I, I, P, P, P, P, P],
    "lego_linear_bwd_data": [P, I, P, I, P, I, I, P, I, I, I, P, I, F, P, P, P, P, P, P],
    "lego_linear_bwd_weight": [P, I, P, I, P, I, I, P, I, I, P, P, P],
    "lego_colsum": [P, I, I, P, P, I, P, P],
    "lego_conv3_pack": [P, P, I, I, P],
    "lego_conv3_unpack_add": [P, P, I, I, P],
    "lego_dropout_mask": [P, I, P, I, P, P],
    "lego_plan_pairs": [P, I, P, P, P, P],
    "lego_conv3_wino_pack": [P, P, P, I, I, P],
    "lego_conv3_wino_unpack_add": [P, I, P, I, I, P],
    "lego_conv3_wino_fwd": [P, I, P, P, P, I, P, P, I, I, I, P, P],
    "lego_conv3_wino_bwd_data": [P, I, P, P, P, I, P, P, I, I, I, P, P, P],
    "lego_conv3_wino_bwd_weight": [P, I, P, I, P, I, P, P, I, I, I, P],
    "lego_conv3_fwd": [P, I, P, P, P, P, I, I, P, I, I, P, I, P],
    "lego_conv3_bwd_data": [P, I, P, P, P, I, I, P, I, I, P, P, I, P],
    "lego_conv3_bwd_weight": [P, I, P, I, P, P, I, P, I, I, P],
    "lego_additive_pool_fwd": [P, I, P, I, P, P, P, P, I, P, I, I, P, I, P, P],
    "lego_additive_pool_bwd_fold": [P, I, P, P, P],
    "lego_additive_pool_bwd": [P, I, P, I, P, P, P, I, P, I, I, P, I, P, P, I, P, P, P, P],
    "lego_dot_ce_fwd": [P, I, P, I, I, I, I, P, P, P],
    "lego_dot_ce_bwd": [P, I, P, I, P, I, I, I, F, P, P, I, P, I, P],
    "lego_mhsa_long_segments": [P, I, P, P, P, P],
    "lego_mhsa_core_fwd": [P, I, P, I, P, I, I, P, I, P, P, I, P, I, I, P, P, P],
    "lego_mhsa_core_bwd": [P, I, P, I, P, I, I, P, I, P, P, I, P, I, P, I, P, I, P, P, P],
    "lego_user_tower_train": [P, I, P, I, P, P, I, I, I, I, I, F, P, P, P, P, I, P, P, P],
    "lego_rowdot_fwd": [P, I, P, I, I, I, P, P],
    "lego_rowdot_bwd": [P, I, P, I, P, I, I, P, I, P, I, P],
    "lego_relu_bwd": [P, I, P, I, I, I, F, P],
    "lego_adam_step": [P, P, P, P, I64, F, F, F, F, I, F, I, P],
    "lego_set_product_mode": [I],
    "lego_adam_step_rows": [P, P, P, P, I, I, P, F, F, F, F, I, F, I, P],
    "lego_mark_rows": [P, I, P, I, P, P],
    "lego_sample_negatives": [P, P, P, P, I, I, I, I, U64, U32, U32, U32, P, P, P],
    "lego_gather_history": [P, P, P, I, I, P, P, P],
    "lego_gather_i32": [P, P, I, P, P, P],
    "lego_segment_live": [P, I, P, P, P],
    "lego_dropout_add_layernorm_fwd": [P, I, P, I, P, P, F, P, P, P, I, P, P, I, I, P],
    "lego_dropout_add_layernorm_bwd": [P, I, P, I, P, I, P, P, P, P, P, P, I, P, I, P, P, P, I, I, P],
    "lego_gelu_fwd": [P, P, I64, P],
    "lego_gelu_bwd": [P, P, P, I64, P],
    "lego_linear_gelu_fwd": [P, I, P, I, P, P, I, P, I, I, I, I, P],
    "lego_linear_bwd_data_gelu": [P, I, P, I, P, I, P, I, I, I, I, P],
    "lego_grouped_metrics": [P, P, P, I, P, I, P, P],
}


# entry points that return a VALUE instead of a status (bound separately; tests/test_abi.py checks them against the header too)
VALUE_FUNCS = {"lego_conv3_wino_du_slabs": [I, I, I], "lego_get_product_mode": []}
VALUE_FUNCS_I64 = {"lego_sort_rows_temp_bytes": [I]}


class LegoHipError(RuntimeError):
    pass


def build(verbose: bool = False) -> str:
    """Compile every HIP source for gfx950 into legommenders_amd/csrc/liblego_hip.so (in-tree)."""
    cmd = ["make", "-C", CSRC, "-j4"]
    res = subprocess.run(cmd, capture_output=True, text=True)
    if verbose or res.returncode != 0:
        print(res.stdout[-4000:])
        print(res.stderr[-4000:])
    if res.returncode != 0:
        raise LegoHipError("building liblego_hip.so failed (hipcc --offload-arch=gfx950)")
    return LIB_PATH


_lib = None


def _single_hip_runtime():
    """PyTorch-ROCm ships its own libamdhip64.so.7; liblego_hip.so must bind to THAT copy (same
    soname), otherwise two HIP runtimes live in one process and torch's streams / device state are
    invisible to our launches.  Loading torch's copy first makes the dynamic loader reuse it."""
    import torch
    cand = os.path.join(os.path.dirname(torch.__file__), "lib", "libamdhip64.so")
    if os.path.exists(cand):
        ctypes.CDLL(cand, mode=ctypes.RTLD_GLOBAL)


def lib() -> ctypes.CDLL:
    """The loaded library; raises (never falls back) if it is absent or incomplete."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise LegoHipError(
            f"{LIB_PATH} not found: the HIP extension is required (no CPU fallback). "
            "Run `python -c 'import __graft_entry__ as g; g.build()'` or `make -C legommenders_amd/csrc`.")
    _single_hip_runtime()
    handle = ctypes.CDLL(LIB_PATH)
    handle.lego_last_error.restype = ctypes.c_char_p
    handle.lego_last_error.argtypes = []
    handle.lego_abi_version.restype = ctypes.c_int
    handle.lego_abi_version.argtypes = []
    got = handle.lego_abi_version()
    if got != ABI_VERSION:                      # a stale in-tree build would be called with shifted argument lists
        raise LegoHipError(f"{LIB_PATH} reports C-ABI version {got}, this binding expects {ABI_VERSION}: "
                           "rebuild with `make -C legommenders_amd/csrc`")
    for name, argtypes in VALUE_FUNCS.items():
        fn = getattr(handle, name)
        fn.restype, fn.argtypes = ctypes.c_int, argtypes
    for name, argtypes in VALUE_FUNCS_I64.items():
        fn = getattr(handle, name)
        fn.restype, fn.argtypes = ctypes.c_int64, argtypes
    for name, argtypes in SIGNATURES.items():
        try:
            fn = getattr(handle, name)
        except AttributeError as exc:
            raise LegoHipError(f"liblego_hip.so does not export {name}") from exc
        fn.restype = ctypes.c_int
        fn.argtypes = argtypes
    _lib = handle
    return handle


EXACT_F32, SPLIT_BF16 = 0, 1


def set_product_mode(mode: int) -> None:
    """process-wide product mode of the large dense products (include/lego_hip.h): EXACT_F32 (default, the parity mode) or
    SPLIT_BF16 (opt-in, ~2x the rate, relative error ~4e-6 per product).  Engines read it when they are BUILT (the NAML engine
    takes the direct conv for the forward and the data gradient in split mode and keeps the exact Winograd weight gradient), so set it
    before constructing a TrainStep / Evaluator."""
    call("lego_set_product_mode", int(mode))


def product_mode() -> int:
    return int(lib().lego_get_product_mode())


def call(name: str, *args) -> None:
    handle = lib()
    rc = getattr(handle, name)(*args)
    if rc != 0:
        raise LegoHipError(f"{name}: {handle.lego_last_error().decode(errors='replace')}")


def _header_text():
    import re
    text = open(HEADER).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return re.sub(r"//[^\n]*", "", text)


def declared_symbols():
    """Function names declared in include/lego_hip.h (used by the CPU-side ABI test)."""
    import re
    return sorted(set(re.findall(r"\b(lego_[a-z0-9_]+)\s*\(", _header_text())))


_C_KINDS = {"int": I, "int32_t": I, "float": F, "int64_t": I64, "uint64_t": U64, "uint32_t": U32}


def declared_prototypes():
    """name -> (restype token, [ctypes kind per parameter]) parsed from include/lego_hip.h: every pointer parameter is
    c_void_p, scalars by their C type.  tests/test_abi.py holds SIGNATURES to this, so a changed argument list in the
    header (or in the binding) fails on the CPU, not as a crash in a GPU test."""
    import re
    out = {}
    for ret, name, args in re.findall(r"\b(int64_t|int|const\s+char\s*\*)\s+(lego_[a-z0-9_]+)\s*\(([^)]*)\)\s*;", _header_text()):
        kinds = []
        for a in (x.strip() for x in args.split(",")):
            if not a or a == "void":
                continue
            if "*" in a:
                kinds.append(P)
                continue
            toks = [t for t in re.split(r"\s+", a) if t not in ("const", "unsigned")]
            if toks[0] not in _C_KINDS:
                raise LegoHipError(f"{name}: cannot classify parameter {a!r}")
            kinds.append(_C_KINDS[toks[0]])
        out[name] = (ret, kinds)
    return out
