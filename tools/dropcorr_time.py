"""lego_qkv_expand_dropcorr alone at the NRMS bench shape (27 k rows, 4.5 k keys, D = 256): with the site's keep bits at p = 0.1, with keep bits
that drop nothing (the memory side of the kernel alone), and without Dropout (plain expansion).
    python tools/dropcorr_time.py"""
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from legommenders_amd._lib import call, LegoDropout  # noqa: E402

dev = torch.device("cuda:0")
D, R, U = 256, 27600, 4500
N = 3 * D


def P(t):
    return None if t is None else ctypes.c_void_p(t.data_ptr())


def bench(fn, n=50):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    a, c = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    c.record()
    torch.cuda.synchronize()
    return a.elapsed_time(c) / n * 1e3


torch.manual_seed(0)
Eu = torch.randn(U, D, device=dev)
W = torch.randn(N, D, device=dev) * 0.05
WT = W.t().contiguous()
b = torch.randn(N, device=dev)
QKVu = Eu @ W.t()
z = torch.distributions.Zipf if False else None
inv = (torch.rand(R, device=dev) ** 3 * U).int().clamp_(0, U - 1).contiguous()          # skewed towards the small keys, like token frequencies
ri = torch.full((R,), 4, dtype=torch.int32, device=dev)
ri[::7] = 0                                                                                 # a seventh of the rows: [SEP] / category positions
cnt = torch.tensor([R], dtype=torch.int32, device=dev)
out = torch.zeros(R, N, device=dev)
mask = torch.zeros(((R + 3) // 4) * D + 4, dtype=torch.uint8, device=dev)
for p in (0.1, 0.2, 0.05):
    call("lego_dropout_mask", ctypes.byref(LegoDropout(p, 1, 1, None)), R, P(cnt), D, P(mask), None)
    dr = ctypes.byref(LegoDropout(p, 1, 1, mask.data_ptr()))
    f = lambda: call("lego_qkv_expand_dropcorr", P(QKVu), N, P(Eu), D, P(WT), N, P(b), P(inv), P(ri), dr, R, P(cnt), D, N, P(out), N, None)
    print(f"keep bits at p={p}: {bench(f):.1f} us", flush=True)
mask.fill_(0xFF)
dr = ctypes.byref(LegoDropout(0.1, 1, 1, mask.data_ptr()))
f = lambda: call("lego_qkv_expand_dropcorr", P(QKVu), N, P(Eu), D, P(WT), N, P(b), P(inv), P(ri), dr, R, P(cnt), D, N, P(out), N, None)
print(f"keep bits that drop nothing: {bench(f):.1f} us")
f = lambda: call("lego_qkv_expand_dropcorr", P(QKVu), N, P(Eu), D, None, N, P(b), P(inv), P(ri), None, R, P(cnt), D, N, P(out), N, None)
print(f"no Dropout (plain expansion): {bench(f):.1f} us")
f = lambda: call("lego_expand_rows", P(QKVu), N, P(inv), R, P(cnt), N, None, None, None, 0, None, None, 0, None, P(out), N, None)
print(f"lego_expand_rows of the same rows: {bench(f):.1f} us")
f = lambda: call("lego_linear_fwd", P(Eu), D, P(W), D, None, P(QKVu), N, U, None, N, D, 0, None, None, None, None, None)
print(f"per-key product [{U} x {D}] x [{D} x {N}]: {bench(f):.1f} us")
Er = torch.randn(R, D, device=dev)
f = lambda: call("lego_linear_fwd", P(Er), D, P(W), D, P(b), P(out), N, R, None, N, D, 0, None, None, None, None, None)
print(f"row-by-row product [{R} x {D}] x [{D} x {N}]: {bench(f):.1f} us")
