import sys, torch, time
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from legommenders_amd import kernels as K
dev = torch.device('cuda:0')
def bench(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
for (M, N, Kd) in [(16384, 256, 768), (32768, 256, 768), (65536, 256, 768), (26368, 256, 768), (32768, 256, 256), (32768, 256, 300), (16384,256,256)]:
    x = torch.randn(M, Kd, device=dev); W = torch.randn(N, Kd, device=dev) * 0.05; b = torch.randn(N, device=dev)
    y = torch.empty(M, N, device=dev)
    ms = bench(lambda: K.linear_fwd(x, W, b, act=0, out=y))
    print(f"NT  M={M} N={N} K={Kd}: {ms*1e3:8.1f} us  {2*M*N*Kd/ms/1e9:7.1f} TF/s")
    g = torch.randn(M, N, device=dev)
    if Kd % 4 == 0:
        dx = torch.empty(M, Kd, device=dev)
        ms = bench(lambda: K.linear_bwd_data(g, W, accumulate_into=None))
        print(f"NN  M={M} N={N} K={Kd}: {ms*1e3:8.1f} us  {2*M*N*Kd/ms/1e9:7.1f} TF/s")
        dW = torch.zeros(N, Kd, device=dev)
        ms = bench(lambda: K.linear_bwd_weight(g, x, dW))
        print(f"TN  M={M} N={N} K={Kd}: {ms*1e3:8.1f} us  {2*M*N*Kd/ms/1e9:7.1f} TF/s")
