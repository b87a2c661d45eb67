"""Winograd conv kernels vs the direct three-tap kernels on the same ragged plan (GPU)."""
import sys, os, ctypes, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from legommenders_amd import _lib
from legommenders_amd._lib import call
from legommenders_amd.engine import LegoDropout
dev = torch.device("cuda:0")
torch.manual_seed(0)
def P(t, off=0): return ctypes.c_void_p(t.data_ptr() + off * t.element_size())
D = int(sys.argv[1]) if len(sys.argv) > 1 else 256
NI0 = int(sys.argv[2]) if len(sys.argv) > 2 else 3500
lens = torch.randint(1, 31, (NI0,), device=dev, dtype=torch.int32)
seg = torch.zeros(NI0 + 1, dtype=torch.int32, device=dev); seg[1:] = torch.cumsum(lens, 0)
R = int(seg[-1]); NI = NI0
pos = torch.arange(R, device=dev) - torch.repeat_interleave(seg[:-1].long(), lens.long())
ln = torch.repeat_interleave(lens.long(), lens.long())
inst = torch.repeat_interleave(torch.arange(NI, device=dev), lens.long())
rowinfo = ((pos > 0).int() | ((pos < ln - 1).int() << 1) | 4 | (inst.int() << 8)).int().contiguous()
cnt = torch.tensor([R, NI, R + NI, 0, 0, 0, 0, 0], dtype=torch.int32, device=dev)
pair = torch.zeros(NI * 15, dtype=torch.int32, device=dev)
call("lego_plan_pairs", P(seg), NI, P(cnt, 1), P(pair), P(cnt, 5), None)
Pn = int(cnt[5]); print("rows", R, "pairs", Pn, "expected", int(((lens + 1) // 2).sum()))
h = torch.randn(R, D, device=dev); w = torch.randn(D, D, 3, device=dev) * 0.05; b = torch.randn(D, device=dev)
wt = torch.zeros(3, D, D, device=dev); u = torch.zeros(4, D, D, device=dev)
call("lego_conv3_pack", P(w), P(wt), D, D, None); ut = torch.zeros(4, D, D, device=dev); call("lego_conv3_wino_pack", P(w), P(u), P(ut), D, D, None)
def bench(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    a, c = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    c.record(); torch.cuda.synchronize()
    return a.elapsed_time(c) / n * 1e3
for p in (0.0, 0.1):
    dr = ctypes.byref(LegoDropout(p, 2023, 5)) if p > 0 else None
    y0 = torch.zeros(R, D, device=dev); y1 = torch.zeros(R, D, device=dev)
    f0 = lambda: call("lego_conv3_fwd", P(h), D, P(wt), P(b), P(rowinfo), P(y0), D, R, P(cnt, 0), D, D, dr, 0, None)
    f1 = lambda: call("lego_conv3_wino_fwd", P(h), D, P(u), P(b), P(pair), pair.numel(), P(cnt, 5), P(y1), D, D, D, dr, None)
    t0, t1 = bench(f0), bench(f1)
    print(f"fwd p={p}: direct {t0:.1f} us  wino {t1:.1f} us  maxdiff {(y0 - y1).abs().max().item():.2e} (scale {y0.abs().max().item():.2f})")
    gy = torch.randn(R, D, device=dev)
    d0 = torch.zeros(R, D, device=dev); d1 = torch.zeros(R, D, device=dev); c0 = torch.zeros(D, device=dev); c1 = torch.zeros(D, device=dev)
    g0 = lambda: call("lego_conv3_bwd_data", P(gy), D, P(wt), P(rowinfo), P(d0), D, R, P(cnt, 0), D, D, dr, P(c0), 0, None)
    g1 = lambda: call("lego_conv3_wino_bwd_data", P(gy), D, P(u), P(ut) if os.environ.get("WINO_UT") else None, P(pair), pair.numel(), P(cnt, 5), P(d1), D, D, D, dr, None if os.environ.get("NO_COLSUM") else P(c1), None)
    t0, t1 = bench(g0), bench(g1)
    print(f"bwd_data p={p}: direct {t0:.1f} us  wino {t1:.1f} us  maxdiff {(d0 - d1).abs().max().item():.2e} (scale {d0.abs().max().item():.2f}) colsum rel {((c0 - c1).abs().max() / c0.abs().max()).item():.2e}")
S = _lib.lib().lego_conv3_wino_du_slabs(D, D, pair.numel())
dwt = torch.zeros(3, D, D, device=dev); du = torch.zeros(S, 4, D, D, device=dev)
gw0 = torch.zeros(D, D, 3, device=dev); gw1 = torch.zeros(D, D, 3, device=dev)
call("lego_conv3_bwd_weight", P(gy), D, P(h), D, P(rowinfo), P(dwt), R, P(cnt, 0), D, D, None)
call("lego_conv3_unpack_add", P(dwt), P(gw0), D, D, None)
call("lego_conv3_wino_bwd_weight", P(gy), D, P(h), D, P(pair), pair.numel(), P(cnt, 5), P(du), D, D, None)
call("lego_conv3_wino_unpack_add", P(du), S, P(gw1), D, D, None)
print(f"bwd_weight maxdiff {(gw0 - gw1).abs().max().item():.2e} (scale {gw0.abs().max().item():.2f})")
t0 = bench(lambda: call("lego_conv3_bwd_weight", P(gy), D, P(h), D, P(rowinfo), P(dwt), R, P(cnt, 0), D, D, None))
t1 = bench(lambda: call("lego_conv3_wino_bwd_weight", P(gy), D, P(h), D, P(pair), pair.numel(), P(cnt, 5), P(du), D, D, None))
print(f"bwd_weight: direct {t0:.1f} us  wino {t1:.1f} us")
