"""Duration and HBM rate of the SHORT-segment attention launches (part = 1) alone at the NRMS item-side shape of a bench step:
1 500 segments of U[8, 33) rows, D = 256, 8 heads, dropout 0.1 (GPU).  Algorithmic bytes: forward reads Q/K/V rows and writes the
output rows + one log-sum-exp per (row, head); backward reads Q/K/V, d(out), lse and writes d(qkv).
    python tools/mhsa_bulk_probe.py"""
import os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from legommenders_amd._lib import call
from legommenders_amd.kernels import _ptr, _stream, _drop
dev = torch.device("cuda:0")
D, heads, n, Lmax = 256, 8, 1500, 33
rs = np.random.RandomState(0)
lens = rs.randint(8, 33, size=n)
seg = torch.tensor(np.concatenate([[0], np.cumsum(lens)]), dtype=torch.int32, device=dev)
R = int(lens.sum())
qkv, go = torch.randn(R, 3 * D, device=dev), torch.randn(R, D, device=dev)
out, gq, lse = torch.empty(R, D, device=dev), torch.empty(R, 3 * D, device=dev), torch.zeros(R, heads, device=dev)
dr = _drop((0.1, 5, 3))
flush = torch.empty(1 << 28, dtype=torch.float32, device=dev)      # 1 GB: evicts the operands from the Infinity Cache between reps


def t(fn, reps=20, cold=False):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    tot = 0.0
    for _ in range(reps):
        if cold:
            flush.add_(1.0)
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record()
        torch.cuda.synchronize()
        tot += a.elapsed_time(b)
    return tot / reps * 1e3


probs_new = torch.zeros(R, heads, Lmax, device=dev)
for mode in ("recompute", "saved"):
    rc = mode == "recompute"
    pb = 0.0 if rc else R * heads * 21 * 4.0
    fb = R * (3 * D + D + heads) * 4.0 + pb
    bb = R * (3 * D + D + heads + 3 * D) * 4.0 + pb
    a_l, a_p = (_ptr(lse), None) if rc else (None, _ptr(probs_new))
    for cold in (False, True):
        f = t(lambda: call("lego_mhsa_core_fwd", _ptr(qkv), 3 * D, _ptr(seg), n, None, D, heads, _ptr(out), D, a_l, a_p, Lmax, dr, R, 1, None, None, _stream()), cold=cold)
        b = t(lambda: call("lego_mhsa_core_bwd", _ptr(qkv), 3 * D, _ptr(seg), n, None, D, heads, _ptr(go), D, a_l, a_p, Lmax, dr, R, _ptr(gq), 3 * D, None, 1, None, None, _stream()), cold=cold)
        print(f"round-4 kernels, {mode:9s} {'cold' if cold else 'warm'} rows={R} fwd {f:6.1f} us ({fb / f / 1e3:6.0f} GB/s)  bwd {b:6.1f} us ({bb / b / 1e3:6.0f} GB/s)")
fb = R * (3 * D + D + heads) * 4.0
bb = R * (3 * D + D + heads + 3 * D) * 4.0

# ---- same box, round-3 kernels (probabilities saved by the forward pass, read by the backward pass): liblego_hip_r03.so when present
old_path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "legommenders_amd", "csrc", "liblego_hip_r03.so")
if os.path.exists(old_path):
    import ctypes
    old = ctypes.CDLL(old_path)
    P, I = ctypes.c_void_p, ctypes.c_int
    old.lego_mhsa_core_fwd.argtypes = [P, I, P, I, P, I, I, P, I, P, I, P, I, I, P, P, P]
    old.lego_mhsa_core_bwd.argtypes = [P, I, P, I, P, I, I, P, I, P, I, P, I, P, I, P, I, P, P, P]
    probs = torch.zeros(R, heads, Lmax, device=dev)
    fb3, bb3 = fb + R * heads * 21 * 4.0, bb + R * heads * 21 * 4.0
    for cold in (False, True):
        f = t(lambda: old.lego_mhsa_core_fwd(_ptr(qkv), 3 * D, _ptr(seg), n, None, D, heads, _ptr(out), D, _ptr(probs), Lmax, dr, R, 1, None, None, _stream()), cold=cold)
        b = t(lambda: old.lego_mhsa_core_bwd(_ptr(qkv), 3 * D, _ptr(seg), n, None, D, heads, _ptr(go), D, _ptr(probs), Lmax, dr, R, _ptr(gq), 3 * D, None, 1, None, None, _stream()), cold=cold)
        print(f"round-3 kernels (saved probabilities) {'cold' if cold else 'warm'} fwd {f:6.1f} us ({fb3 / f / 1e3:6.0f} GB/s)  bwd {b:6.1f} us ({bb3 / b / 1e3:6.0f} GB/s)")
