"""Timings of the two plain-row strip products of the NAML step at the bench shape (27.6 k rows, 256 x 256): the additive hidden layer
(NT, tanh epilogue) and its data gradient (NN, accumulate + ReLU-backward epilogue).  With the tuning library and LEGO_DMA_ABL=<bits>
(1 no DMA in the k loop, 4 no wait / barrier, 8 no epilogue, 32 no MFMAs) it is the ablation table of gemm_dma.hpp.
    LEGO_HIP_LIB=.../liblego_hip_tune.so LEGO_DMA_ABL=8 python tools/strip_ablation.py"""
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from legommenders_amd._lib import call  # noqa: E402

dev = torch.device("cuda:0")


def P(t):
    return None if t is None else ctypes.c_void_p(t.data_ptr())


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


R, D, A = int(sys.argv[1]) if len(sys.argv) > 1 else 27613, 256, 256
torch.manual_seed(0)
x = torch.randn(R, D, device=dev)
W = torch.randn(A, D, device=dev) * 0.05
b = torch.randn(A, device=dev)
t = torch.zeros(R, A, device=dev)
cnt = torch.tensor([R], dtype=torch.int32, device=dev)
f = lambda: call("lego_linear_fwd", P(x), D, P(W), D, P(b), P(t), A, R, P(cnt), A, D, 2, None, None, None, None, None)
dy = torch.randn(R, D, device=dev)
cs = torch.zeros(D, device=dev)
g = lambda: call("lego_linear_bwd_data", P(t), A, P(W), D, P(dy), D, R, P(cnt), A, D, 1, P(x), D, 1.1, None, None, P(cs), None, None, None)
tf, tg = bench(f), bench(g)
fl = 2.0 * R * D * A
print(f"LEGO_DMA_ABL={os.environ.get('LEGO_DMA_ABL', '0')}: NT tanh {tf:.1f} us ({fl / tf * 1e-6:.1f} TFLOP/s)   NN accumulate + relu' {tg:.1f} us ({fl / tg * 1e-6:.1f} TFLOP/s)")
