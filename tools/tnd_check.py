"""lego_linear_bwd_weight (whichever kernel LEGO_TND selects: 0 = tile kernels of gemm_tn.hpp, 1 = LDS-free tnd_kernel) against a float64
product, on the path's weight-gradient shapes, with timings.
    LEGO_TND=1 python tools/tnd_check.py [--time-only]"""
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from legommenders_amd._lib import call  # noqa: E402

dev = torch.device("cuda:0")


def P(t, off=0):
    return None if t is None else ctypes.c_void_p(t.data_ptr() + off * t.element_size())


def bench(fn, n=30):
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


def case(R, N, K, cap=None, ldg=None, ldx=None, off=False, timing=False, seed=0):
    """dW[N, K] += g[R, N]^T x[R, K]; `cap` > R: the launch is sized for a capacity and reads the live row count from the device"""
    torch.manual_seed(seed)
    cap = cap or R
    ldg, ldx = ldg or N, ldx or K
    goff = 5 if off else 0
    g = torch.randn(cap + goff + 1, ldg, device=dev)
    x = torch.randn(cap + 1, ldx, device=dev)
    g[goff + R:] = float("nan")                      # rows past the live count must never be read as data
    x[R:] = float("nan")
    dW = torch.randn(N, K, device=dev)
    ref = dW.double() + g[goff:goff + R, :N].double().T @ x[:R, :K].double()
    cnt = torch.tensor([R, goff], dtype=torch.int32, device=dev)
    out = dW.clone()
    f = lambda: call("lego_linear_bwd_weight", P(g), ldg, P(x), ldx, P(out), K, cap, P(cnt, 0), N, K, P(cnt, 1) if off else None, None, None)
    f()
    torch.cuda.synchronize()
    err = (out.double() - ref).abs().max().item()
    scale = ref.abs().max().item()
    ok = err <= 2e-5 * scale
    line = f"R={R} cap={cap} N={N} K={K} ld=({ldg},{ldx}) off={off}: max err {err:.2e} (scale {scale:.1f})"
    if timing:
        t = bench(f)
        line += f" | {t:.1f} us  {2.0 * R * N * K / t * 1e-6:.1f} TFLOP/s"
    print(("ok   " if ok else "FAIL ") + line, flush=True)
    return ok


if __name__ == "__main__":
    print("LEGO_TND =", os.environ.get("LEGO_TND", "(default 1)"))
    good = True
    if "--time-only" not in sys.argv:
        for kw in [dict(R=2048, N=64, K=64), dict(R=2051, N=256, K=300, cap=2600), dict(R=4097, N=256, K=256, cap=105600, off=True),
                   dict(R=3000, N=100, K=36, ldg=104, ldx=40), dict(R=9000, N=768, K=256, ldg=768, ldx=260), dict(R=1, N=256, K=256, cap=4000),
                   dict(R=2500, N=12, K=8, cap=2500), dict(R=27613, N=256, K=256, cap=109120)]:
            good &= case(**kw)
    # the path's shapes: additive hidden layer (token rows + instances), projection over the distinct tokens, NRMS in-projection, BERT FFN
    for kw in [dict(R=27613, N=256, K=256, cap=109120), dict(R=4500, N=256, K=300, cap=105600), dict(R=30700, N=768, K=256, cap=123200),
               dict(R=29600, N=3072, K=768), dict(R=29600, N=768, K=3072), dict(R=29600, N=768, K=768)]:
        good &= case(timing=True, **kw)
    print("ALL OK" if good else "FAILED")
    sys.exit(0 if good else 1)
