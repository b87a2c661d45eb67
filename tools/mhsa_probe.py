"""The attention core alone at the NRMS item-side shape of a bench step (1 500 segments of U[8, 33) rows, D = 256, 8 heads, dropout 0.1;
saved-probabilities mode = the engine's), warm and cold caches; and at BERT's head dim 64 (12 heads, 1 850 segments).  With
LEGO_HIP_LIB pointing at another build the same script times that build: `tools/r06/gpu4.sh` runs both on one box.
    python tools/mhsa_probe.py"""
import os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from legommenders_amd._lib import call, LIB_PATH
from legommenders_amd.kernels import _ptr, _stream, _drop
dev = torch.device("cuda:0")
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


for name, D, heads, n, lo, hi in (("nrms item side", 256, 8, 1500, 8, 33), ("short titles only", 256, 8, 1500, 8, 17), ("bert head dim 64", 768, 12, 1850, 8, 33)):
    rs = np.random.RandomState(0)
    lens = rs.randint(lo, hi, size=n)
    Lmax = 33
    seg = torch.tensor(np.concatenate([[0], np.cumsum(lens)]), dtype=torch.int32, device=dev)
    R = int(lens.sum())
    qkv, go = torch.randn(R, 3 * D, device=dev), torch.randn(R, D, device=dev)
    out, gq = torch.empty(R, D, device=dev), torch.empty(R, 3 * D, device=dev)
    probs = torch.zeros(R, heads, Lmax, device=dev)
    dr = _drop((0.1, 5, 3))
    for cold in (False, True):
        f = t(lambda: call("lego_mhsa_core_fwd", _ptr(qkv), 3 * D, _ptr(seg), n, None, D, heads, _ptr(out), D, None, _ptr(probs), Lmax, dr, R, 0, None, None, _stream()), cold=cold)
        b = t(lambda: call("lego_mhsa_core_bwd", _ptr(qkv), 3 * D, _ptr(seg), n, None, D, heads, _ptr(go), D, None, _ptr(probs), Lmax, dr, R, _ptr(gq), 3 * D, None, 0, None, None, _stream()), cold=cold)
        print(f"{os.path.basename(LIB_PATH):26s} {name:18s} {'cold' if cold else 'warm'} rows={R} share<=16: {float((lens <= 16).mean()):.2f}  fwd {f:6.1f} us  bwd {b:6.1f} us", flush=True)
