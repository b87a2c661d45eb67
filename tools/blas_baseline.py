"""Vendor-library reference point for the path's GEMM shapes: torch.matmul (hipBLASLt / rocBLAS, fp32) timed with HIP events.
Not part of the product -- a yardstick for the hand-written strip / tn kernels."""
import torch

def t(fn, n=30):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3

torch.backends.cuda.matmul.allow_tf32 = False
dev = torch.device("cuda:0")
shapes = [("nrms qkv fwd       NT", 30720, 768, 256), ("nrms out_proj fwd  NT", 30720, 256, 256), ("naml proj fwd      NT", 26145, 256, 300),
          ("naml additive fwd  NT", 26145, 200, 256), ("naml conv as GEMM  NT", 26145, 256, 768)]
for name, M, N, K in shapes:
    x = torch.randn(M, K, device=dev); w = torch.randn(N, K, device=dev)
    us = t(lambda: torch.matmul(x, w.t()))
    print(f"{name}  M={M} N={N} K={K}: {us:7.1f} us  {2*M*N*K/us*1e-6:6.1f} TFLOP/s")
    g = torch.randn(M, N, device=dev)
    us = t(lambda: torch.matmul(g, w))                      # bwd data NN
    print(f"{'   bwd data        NN':22s}  M={M} N={K} K={N}: {us:7.1f} us  {2*M*N*K/us*1e-6:6.1f} TFLOP/s")
    us = t(lambda: torch.matmul(g.t(), x))                  # bwd weight TN
    print(f"{'   bwd weight      TN':22s}  M={N} N={K} K={M}: {us:7.1f} us  {2*M*N*K/us*1e-6:6.1f} TFLOP/s")
