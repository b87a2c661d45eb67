"""lego_linear_fwd on the shapes rows2_kernel takes (whichever kernel LEGO_ROWS2 selects) against float64, + timings.
    LEGO_ROWS2=1 python tools/rows2_check.py"""
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from legommenders_amd._lib import call  # noqa: E402

dev = torch.device("cuda:0")


def P(t):
    return None if t is None else ctypes.c_void_p(t.data_ptr())


def bench(fn, n=40):
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


def case(R, N, K, cap=None, act=0, bias=True, ldx=None, ldo=None, timing=False, seed=0):
    torch.manual_seed(seed)
    cap = cap or R
    ldx, ldo = ldx or K, ldo or N
    x = torch.randn(cap + 1, ldx, device=dev)
    x[R:] = float("nan")                                 # rows past the live count must not reach a live output
    W = torch.randn(N, K, device=dev) * 0.05
    b = torch.randn(N, device=dev) if bias else None
    out = torch.full((cap + 1, ldo), 7.0, device=dev)
    cnt = torch.tensor([R], dtype=torch.int32, device=dev)
    f = lambda: call("lego_linear_fwd", P(x), ldx, P(W), K, P(b), P(out), ldo, cap, P(cnt), N, K, act, None, None, None, None, None)
    f()
    torch.cuda.synchronize()
    ref = x[:R, :K].double() @ W.double().T + (b.double() if bias else 0.0)
    if act == 1:
        ref = ref.clamp_min(0)
    elif act == 2:
        ref = torch.tanh(ref)
    err = (out[:R, :N].double() - ref).abs().max().item()
    untouched = float(out[R:].min()) == 7.0 == float(out[R:].max()) and (ldo == N or (float(out[:, N:].min()) == 7.0 == float(out[:, N:].max())))
    ok = err <= 3e-5 * max(1.0, ref.abs().max().item()) and untouched
    line = f"R={R} cap={cap} N={N} K={K} act={act} ld=({ldx},{ldo}): max err {err:.2e}, rows / columns outside the product untouched: {untouched}"
    if timing:
        t = bench(f)
        line += f" | {t:.1f} us  {2.0 * R * N * K / t * 1e-6:.1f} TFLOP/s"
    print(("ok   " if ok else "FAIL ") + line, flush=True)
    return ok


def case_nn(R, N, K, cap=None, accumulate=1, relu=True, colsum=True, timing=False, seed=0):
    """dx[R, K] (+)= g[R, N] . W[N, K]  [masked by relu_ref > 0, scaled] + column sums: lego_linear_bwd_data"""
    torch.manual_seed(seed)
    cap = cap or R
    g = torch.randn(cap + 1, N, device=dev)
    g[R:] = float("nan")
    W = torch.randn(N, K, device=dev) * 0.05
    dx0 = torch.randn(cap + 1, K, device=dev)
    ref_t = torch.randn(cap + 1, K, device=dev)
    cs = torch.zeros(K, device=dev)
    cnt = torch.tensor([R], dtype=torch.int32, device=dev)
    dx = dx0.clone()
    f = lambda: call("lego_linear_bwd_data", P(g), N, P(W), K, P(dx), K, cap, P(cnt), N, K, accumulate, P(ref_t) if relu else None, K, 1.25 if relu else 1.0,
                     None, None, P(cs) if colsum else None, None, None, None)
    f()
    torch.cuda.synchronize()
    want = g[:R].double() @ W.double()
    if accumulate:
        want = want + dx0[:R].double()
    if relu:
        want = torch.where(ref_t[:R] > 0, 1.25 * want, torch.zeros_like(want))
    err = (dx[:R].double() - want).abs().max().item()
    ecs = (cs.double() - want.sum(0)).abs().max().item() / max(1.0, want.sum(0).abs().max().item()) if colsum else 0.0
    untouched = bool(torch.equal(dx[R:], dx0[R:]))
    ok = err <= 3e-5 * max(1.0, want.abs().max().item()) and ecs < 1e-4 and untouched
    line = f"NN R={R} cap={cap} N={N} K={K} accumulate={accumulate} relu={relu}: max err {err:.2e}, colsum rel {ecs:.1e}, rows past the count untouched: {untouched}"
    if timing:
        t = bench(f)
        line += f" | {t:.1f} us  {2.0 * R * N * K / t * 1e-6:.1f} TFLOP/s"
    print(("ok   " if ok else "FAIL ") + line, flush=True)
    return ok


if __name__ == "__main__":
    print("LEGO_ROWS2 =", os.environ.get("LEGO_ROWS2", "(default 1)"))
    good = True
    for kw in [dict(R=8200, N=256, K=256, act=2), dict(R=9001, N=128, K=64, cap=9100, act=1), dict(R=1, N=256, K=256, cap=9000), dict(R=8500, N=260, K=96, ldo=264, bias=False),
               dict(R=10000, N=768, K=256, cap=30000, ldx=260), dict(R=27613, N=256, K=256, cap=109120, act=2), dict(R=113, N=256, K=32, cap=8192),
               dict(R=9000, N=256, K=300, cap=9100), dict(R=8200, N=64, K=20), dict(R=8200, N=256, K=4, act=1)]:
        good &= case(**kw)
    for kw in [dict(R=27613, N=256, K=256, cap=109120, act=2), dict(R=4500, N=256, K=300, cap=105600), dict(R=30700, N=768, K=256, cap=123200), dict(R=30700, N=256, K=256, cap=123200),
               dict(R=29600, N=768, K=768), dict(R=29600, N=3072, K=768), dict(R=29600, N=768, K=3072)]:
        good &= case(timing=True, **kw)
    for kw in [dict(R=8200, N=256, K=256), dict(R=9001, N=64, K=128, cap=9100, relu=False), dict(R=1, N=256, K=256, cap=9000, accumulate=0, relu=False, colsum=False),
               dict(R=8500, N=96, K=260, relu=False, colsum=False), dict(R=113, N=32, K=256, cap=8192), dict(R=9000, N=300, K=256, cap=9100), dict(R=8200, N=12, K=64)]:
        good &= case_nn(**kw)
    for kw in [dict(R=27613, N=256, K=256, cap=109120), dict(R=30700, N=256, K=256, cap=123200, relu=False, colsum=False),
               dict(R=30700, N=768, K=256, cap=123200, accumulate=0, relu=False, colsum=False)]:
        good &= case_nn(timing=True, **kw)
    print("ALL OK" if good else "FAILED")
    sys.exit(0 if good else 1)
