"""How the attention core's launch time grows with the number of (segment, head) pairs: fixed segment length, n segments, 8 heads of 32, warm caches.
The slope is the per-pair cost of a SIMD's share, the intercept the launch's fixed part.    python tools/mhsa_scaling.py"""
import os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from legommenders_amd._lib import call
from legommenders_amd.kernels import _ptr, _stream, _drop
dev = torch.device("cuda:0")


def t(fn, reps=30):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3


D, heads, Lmax = 256, 8, 33
for L in (32, 24, 16, 8):
    for p in (0.1, 0.0):
        row = []
        for n in (128, 512, 1024, 1536, 3072, 6144):
            seg = torch.arange(0, (n + 1) * L, L, dtype=torch.int32, device=dev)
            R = n * L
            qkv, go = torch.randn(R, 3 * D, device=dev), torch.randn(R, D, device=dev)
            out, gq = torch.empty(R, D, device=dev), torch.empty(R, 3 * D, device=dev)
            probs = torch.zeros(R, heads, Lmax, device=dev)
            dr = _drop((p, 5, 3)) if p > 0 else None
            f = t(lambda: call("lego_mhsa_core_fwd", _ptr(qkv), 3 * D, _ptr(seg), n, None, D, heads, _ptr(out), D, None, _ptr(probs), Lmax, dr, R, 0, None, None, _stream()))
            b = t(lambda: call("lego_mhsa_core_bwd", _ptr(qkv), 3 * D, _ptr(seg), n, None, D, heads, _ptr(go), D, None, _ptr(probs), Lmax, dr, R, _ptr(gq), 3 * D, None, 0, None, None, _stream()))
            row.append((n * heads, round(f, 1), round(b, 1)))
        print(f"L={L:2d} p={p}: (pairs, fwd us, bwd us) " + "  ".join(f"{a}:{x}/{y}" for a, x, y in row), flush=True)
