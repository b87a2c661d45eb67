import sys, os, ctypes, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
# tile-variant hooks live in the tuning build only: `make -C legommenders_amd/csrc tune`
os.environ.setdefault('LEGO_HIP_LIB', os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'legommenders_amd', 'csrc', 'liblego_hip_tune.so'))
from legommenders_amd import _lib
L = _lib.lib()
P, I = ctypes.c_void_p, ctypes.c_int
L.lego_debug_gemm_nt.argtypes = [I, P, P, P, P, I, I, I, P]
L.lego_debug_gemm_nt.restype = I
dev = torch.device('cuda:0')
VARIANTS = tuple(int(v) for v in sys.argv[1].split(',')) if len(sys.argv) > 1 else (9, 10)
_blocker = torch.zeros(1 << 27, dtype=torch.float32, device=dev)
def bench(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    for _ in range(3): _blocker.add_(1.0)      # ~1 ms of queued work: the host enqueues the timed launches while it runs
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
names = {0: "128x128 4w(2x2)", 1: "128x128 8w(2x4)", 2: "256x128 8w(4x2)", 3: "128x256 8w(2x4)", 4: "64x128 4w(1x4)", 5: "128x128 8w(4x2)", 6: "64x256 4w(1x4)", 7: "128x128 8w(4x2) stag", 8: "128x128 8w(2x4) stag", 9: "strip 16x16x4", 10: "strip LDS-DMA", 13: "LDS-DMA 4 waves",
         20: "SPLIT 128x128 2x4 stag", 21: "SPLIT 256x128 4x2", 22: "SPLIT 128x256 2x4", 23: "SPLIT 128x128 2x2", 24: "SPLIT 256x128 4x2 stag",
         25: "SPLIT 64x128 1x4", 26: "SPLIT 128x128 4x2 stag"}
SHAPES = [(26368, 256, 768), (24000, 256, 768), (28672, 256, 768), (32768, 256, 768), (65536, 256, 768), (26368, 256, 256), (26368, 256, 300), (26368, 200, 256), (1000, 256, 768), (26368+5, 240, 96), (3200, 200, 256), (3200, 256, 256), (3520, 256, 256), (6400, 256, 256), (26368, 768, 256), (30720, 768, 300), (105600, 256, 256), (26368, 256, 32), (26368, 256, 4), (26368, 256, 36)]
if len(sys.argv) > 2 and sys.argv[2] == 'small':
    SHAPES = [(4530, 256, 300), (4530, 256, 256), (4530, 768, 256), (8000, 256, 300)]
if len(sys.argv) > 2 and sys.argv[2] == 'bert':      # BERT-base block products at the rows of a B = 64 step (config 5)
    SHAPES = [(29600, 3072, 768), (29600, 768, 3072), (29600, 768, 768), (29600, 2304, 768)]
if len(sys.argv) > 2 and sys.argv[2] == 'short':
    SHAPES = [(26368, 256, 768), (26368, 256, 256), (26368, 256, 300), (30720, 768, 256), (105600, 256, 256), (6400, 256, 256)]
for (M, N, Kd) in SHAPES:
    x = torch.randn(M, Kd, device=dev); W = torch.randn(N, Kd, device=dev) * 0.05; b = torch.randn(N, device=dev)
    ref = (x.double() @ W.double().T + b.double())
    for v in VARIANTS:
        y = torch.zeros(M, N, device=dev)
        def run():
            rc = L.lego_debug_gemm_nt(v, x.data_ptr(), W.data_ptr(), b.data_ptr(), y.data_ptr(), M, N, Kd, None)
            assert rc == 0, L.lego_last_error()
        ms = bench(run)
        err = (y.double() - ref).abs().max().item() / ref.abs().max().item()
        print(f"M={M} K={Kd} v{v} {names[v]:22s}: {ms*1e3:8.1f} us {2*M*N*Kd/ms/1e9:7.1f} TF/s  relerr {err:.1e}")
