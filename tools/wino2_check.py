"""Winograd conv kernels (whichever LEGO_WINO2 selects: 0 = wino_kernel, 1 = wino2_kernel staggered, 2 = wino2_kernel plain) against the
direct three-tap kernels on the same ragged plan, WITH the keep bits precomputed as the training step has them (lego_dropout_mask),
at several widths / plan sizes, plus timings at the bench shape.
    LEGO_WINO2=1 python tools/wino2_check.py [--time-only]"""
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from legommenders_amd import _lib  # noqa: E402
from legommenders_amd._lib import call  # noqa: E402
from legommenders_amd.engine import LegoDropout  # noqa: E402

dev = torch.device("cuda:0")


def P(t, off=0):
    return ctypes.c_void_p(t.data_ptr() + off * t.element_size())


def plan(NI, lo, hi, seed):
    g = torch.Generator(device="cpu").manual_seed(seed)
    lens = torch.randint(lo, hi + 1, (NI,), generator=g, dtype=torch.int32).to(dev)
    seg = torch.zeros(NI + 1, dtype=torch.int32, device=dev)
    seg[1:] = torch.cumsum(lens, 0)
    R = int(seg[-1])
    pos = torch.arange(R, device=dev) - torch.repeat_interleave(seg[:-1].long(), lens.long())
    ln = torch.repeat_interleave(lens.long(), lens.long())
    inst = torch.repeat_interleave(torch.arange(NI, device=dev), lens.long())
    rowinfo = ((pos > 0).int() | ((pos < ln - 1).int() << 1) | 4 | (inst.int() << 8)).int().contiguous()
    cnt = torch.tensor([R, NI, R + NI, 0, 0, 0, 0, 0], dtype=torch.int32, device=dev)
    pair = torch.zeros(NI * ((hi + 1) // 2) + 8, dtype=torch.int32, device=dev)
    call("lego_plan_pairs", P(seg), NI, P(cnt, 1), P(pair), P(cnt, 5), None)
    assert int(cnt[5]) == int(((lens + 1) // 2).sum())
    return R, rowinfo, cnt, pair


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


def case(D, NI, lo, hi, seed, p, timing=False):
    torch.manual_seed(seed)
    R, rowinfo, cnt, pair = plan(NI, lo, hi, seed)
    h = torch.randn(R + 2, D, device=dev)[:R]          # two spare rows behind: nothing may read them
    w = torch.randn(D, D, 3, device=dev) * 0.05
    b = torch.randn(D, device=dev)
    wt = torch.zeros(3, D, D, device=dev)
    u = torch.zeros(4, D, D, device=dev)
    ut = torch.zeros(4, D, D, device=dev)
    call("lego_conv3_pack", P(w), P(wt), D, D, None)
    call("lego_conv3_wino_pack", P(w), P(u), P(ut), D, D, None)
    mask = torch.zeros(((R + 3) // 4) * D + 4, dtype=torch.uint8, device=dev)
    dr = None
    if p > 0:
        d0 = LegoDropout(p, 2023, 5, None)
        call("lego_dropout_mask", ctypes.byref(d0), R, P(cnt, 0), D, P(mask), None)
        dr = ctypes.byref(LegoDropout(p, 2023, 5, mask.data_ptr()))
    y0 = torch.zeros(R, D, device=dev)
    y1 = torch.full((R + 2, D), 7.0, device=dev)
    f0 = lambda: call("lego_conv3_fwd", P(h), D, P(wt), P(b), P(rowinfo), P(y0), D, R, P(cnt, 0), D, D, dr, 0, None)
    f1 = lambda: call("lego_conv3_wino_fwd", P(h), D, P(u), P(b), P(pair), pair.numel(), P(cnt, 5), P(y1), D, D, D, dr, None)
    f0(); f1()
    scale = y0.abs().max().item()
    e_f = (y0 - y1[:R]).abs().max().item()
    assert float(y1[R:].min()) == 7.0 and float(y1[R:].max()) == 7.0, "rows past the plan were written"
    gy = torch.randn(R, D, device=dev)
    d0_ = torch.zeros(R, D, device=dev); d1_ = torch.full((R + 2, D), 7.0, device=dev)
    c0 = torch.zeros(D, device=dev); c1 = torch.zeros(D, device=dev)
    g0 = lambda: call("lego_conv3_bwd_data", P(gy), D, P(wt), P(rowinfo), P(d0_), D, R, P(cnt, 0), D, D, dr, P(c0), 0, None)
    g1 = lambda: call("lego_conv3_wino_bwd_data", P(gy), D, P(u), P(ut), P(pair), pair.numel(), P(cnt, 5), P(d1_), D, D, D, dr, P(c1), None)
    g0(); g1()
    e_b = (d0_ - d1_[:R]).abs().max().item()
    sb = d0_.abs().max().item()
    e_c = ((c0 - c1).abs().max() / c0.abs().max()).item()
    # weight gradient: Winograd form over the pairs (slabs + unpack) against the direct three taps
    S = _lib.lib().lego_conv3_wino_du_slabs(D, D, pair.numel())
    dwt = torch.zeros(3, D, D, device=dev)
    du = torch.full((S, 4, D, D), float("nan") if S > 1 else 0.0, device=dev)
    gw0 = torch.zeros(D, D, 3, device=dev); gw1 = torch.zeros(D, D, 3, device=dev)
    w0 = lambda: call("lego_conv3_bwd_weight", P(gy), D, P(h), D, P(rowinfo), P(dwt), R, P(cnt, 0), D, D, None)
    w1 = lambda: call("lego_conv3_wino_bwd_weight", P(gy), D, P(h), D, P(pair), pair.numel(), P(cnt, 5), P(du), D, D, None)
    w0(); call("lego_conv3_unpack_add", P(dwt), P(gw0), D, D, None)
    w1(); call("lego_conv3_wino_unpack_add", P(du), S, P(gw1), D, D, None)
    e_w = (gw0 - gw1).abs().max().item()
    sw = gw0.abs().max().item()
    ok = e_f <= 3e-5 * max(scale, 1) and e_b <= 3e-5 * max(sb, 1) and e_c < 1e-4 and e_w <= 3e-5 * max(sw, 1)
    line = (f"D={D} NI={NI} len {lo}-{hi} rows {R} pairs {int(cnt[5])} p={p}: fwd {e_f:.2e} (scale {scale:.1f}) bwd {e_b:.2e} (scale {sb:.1f}) "
            f"colsum rel {e_c:.1e} wgrad {e_w:.2e} (scale {sw:.1f}, {S} slabs)")
    if timing:
        line += f" | fwd direct {bench(f0):.1f} us wino {bench(f1):.1f} us | bwd direct {bench(g0):.1f} us wino {bench(g1):.1f} us"
        if S > 1:                                    # (a single slab is an atomics accumulator: it would have to be cleared between launches)
            u1 = lambda: call("lego_conv3_wino_unpack_add", P(du), S, P(gw1), D, D, None)
            line += f" | wgrad direct {bench(w0):.1f} us wino {bench(w1):.1f} us + unpack {bench(u1):.1f} us"
    print(("ok   " if ok else "FAIL ") + line, flush=True)
    return ok


if __name__ == "__main__":
    print("LEGO_WINO2 =", os.environ.get("LEGO_WINO2", "(default 1)"))
    good = True
    if "--time-only" not in sys.argv:
        for D, NI, lo, hi, seed, p in [(64, 40, 1, 30, 1, 0.0), (64, 700, 1, 30, 2, 0.1), (128, 300, 1, 9, 3, 0.1), (256, 5, 1, 3, 4, 0.1),
                                       (256, 1, 1, 1, 5, 0.0), (256, 1, 30, 30, 6, 0.1), (256, 333, 1, 30, 7, 0.0), (256, 2100, 1, 30, 8, 0.1),
                                       (96, 900, 2, 30, 9, 0.1), (256, 4000, 29, 30, 10, 0.1)]:
            good &= case(D, NI, lo, hi, seed, p)
    for p in (0.0, 0.1):
        good &= case(256, 1500, 5, 30, 11, p, timing=True)           # ~26 k rows: the bench batch
    print("ALL OK" if good else "FAILED")
    sys.exit(0 if good else 1)
