import sys, os, torch
sys.path.insert(0, os.getcwd())
from legommenders_amd import kernels as K
dev = torch.device("cuda:0")
def bench(fn, n=30):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3
for M in (26368, 50000):
    for (N, Kd) in ((256, 256), (256, 768), (768, 256)):
        x = torch.randn(M, Kd, device=dev); W = torch.randn(N, Kd, device=dev) * 0.05
        g = torch.randn(M, N, device=dev)
        t_nt = bench(lambda: K.linear_fwd(x, W, None, act=0))             # [M,K] x [N,K]^T -> [M,N]
        t_nn = bench(lambda: K.linear_bwd_data(g, W))                     # [M,N] x [N,K]   -> [M,K]
        Wt = W.t().contiguous()                                           # [K,N]: the same product as NT
        t_nn_as_nt = bench(lambda: K.linear_fwd(g, Wt, None, act=0))
        fl = 2.0 * M * N * Kd
        print(f"M={M} N={N} K={Kd}: NT fwd {t_nt:6.1f} us ({fl/t_nt/1e6:5.1f} TF/s) | NN bwd_data {t_nn:6.1f} us ({fl/t_nn/1e6:5.1f}) | same via W^T as NT {t_nn_as_nt:6.1f} us ({fl/t_nn_as_nt/1e6:5.1f})")
