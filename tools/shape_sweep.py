"""The strip kernels on the path's awkward shapes (K or N = 200 / 300) beside round ones, isolated launches with HIP events."""
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
M = 26145
print("NT fwd  [M,K] x [N,K]^T")
for (N, Kd) in ((256, 256), (200, 256), (192, 256), (224, 256), (256, 300), (256, 288), (256, 320), (768, 256)):
    x = torch.randn(M, Kd, device=dev); W = torch.randn(N, Kd, device=dev) * 0.05; b = torch.randn(N, device=dev)
    for act in (0, 2):
        t = bench(lambda: K.linear_fwd(x, W, b, act=act)); fl = 2.0 * M * N * Kd
        print(f"  N={N:4d} K={Kd:4d} act={act}: {t:6.1f} us  {fl/t/1e6:6.1f} TF/s")
print("NN bwd data  [M,N] x [N,K]")
for (N, Kd) in ((256, 256), (200, 256), (192, 256), (224, 256), (768, 256)):
    g = torch.randn(M, N, device=dev); W = torch.randn(N, Kd, device=dev) * 0.05
    t = bench(lambda: K.linear_bwd_data(g, W)); fl = 2.0 * M * N * Kd
    print(f"  red N={N:4d} out K={Kd:4d}: {t:6.1f} us  {fl/t/1e6:6.1f} TF/s")
