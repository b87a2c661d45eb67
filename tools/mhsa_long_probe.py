"""Duration of the LONG-segment attention launches (part = 2) alone, for k segments of 33 rows among 1 500 (GPU)."""
import os, sys, ctypes, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from legommenders_amd._lib import call
from legommenders_amd.kernels import _ptr, _stream, _drop
dev = torch.device("cuda:0")
D, heads, n, Lmax = 256, 8, 1500, 33
rs = np.random.RandomState(0)
for k in (1, 8, 60, 480, 1500):
    lens = rs.randint(8, 33, size=n); lens[rs.choice(n, size=k, replace=False)] = 33
    seg = torch.tensor(np.concatenate([[0], np.cumsum(lens)]), dtype=torch.int32, device=dev)
    R = int(lens.sum())
    qkv, go = torch.randn(R, 3 * D, device=dev), torch.randn(R, D, device=dev)
    out, gq, lse = torch.empty(R, D, device=dev), torch.empty(R, 3 * D, device=dev), torch.zeros(R, heads, device=dev)
    ll, lc = torch.zeros(n, dtype=torch.int32, device=dev), torch.zeros(1, dtype=torch.int32, device=dev)
    call("lego_mhsa_long_segments", _ptr(seg), n, None, _ptr(ll), _ptr(lc), _stream())
    dr = _drop((0.1, 5, 3))
    def t(fn, reps=20):
        for _ in range(3): fn()
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(reps): fn()
        b.record(); torch.cuda.synchronize()
        return a.elapsed_time(b) / reps * 1e3
    for part, name in ((2, "long"), (1, "short")):
        for lst in ((None, None), (_ptr(ll), _ptr(lc))):
            if part == 1 and lst[0] is not None: continue
            f = t(lambda: call("lego_mhsa_core_fwd", _ptr(qkv), 3 * D, _ptr(seg), n, None, D, heads, _ptr(out), D, _ptr(lse), None, Lmax, dr, R, part, *lst, _stream()))
            b = t(lambda: call("lego_mhsa_core_bwd", _ptr(qkv), 3 * D, _ptr(seg), n, None, D, heads, _ptr(go), D, _ptr(lse), None, Lmax, dr, R, _ptr(gq), 3 * D, None, part, *lst, _stream()))
            print(f"k={k:5d} {name:5s} {'listed' if lst[0] is not None else 'ballot':6s} fwd {f:7.1f} us  bwd {b:7.1f} us")
