import sys, os, ctypes, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from legommenders_amd import _lib
L = _lib.lib()
P, I = ctypes.c_void_p, ctypes.c_int
L.lego_debug_gemm_nt.argtypes = [I, P, P, P, P, I, I, I, P]
dev = torch.device('cuda:0')
def bench(fn, n=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
for (M, N, Kd) in [(16384, 256, 256), (16384, 256, 1024), (16384, 256, 4096), (32768, 256, 4096), (32768, 256, 256), (32768,256,1024)]:
    x = torch.randn(M, Kd, device=dev); W = torch.randn(N, Kd, device=dev) * 0.05; b = torch.randn(N, device=dev)
    for v in (0, 5):
        y = torch.zeros(M, N, device=dev)
        ms = bench(lambda: L.lego_debug_gemm_nt(v, x.data_ptr(), W.data_ptr(), b.data_ptr(), y.data_ptr(), M, N, Kd, None))
        print(f"M={M} K={Kd} v{v}: {ms*1e3:8.1f} us {2*M*N*Kd/ms/1e9:7.1f} TF/s")
