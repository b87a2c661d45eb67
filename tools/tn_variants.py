import sys, os, ctypes, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
# tile-variant hooks live in the tuning build only: `make -C legommenders_amd/csrc tune`
os.environ.setdefault('LEGO_HIP_LIB', os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'legommenders_amd', 'csrc', 'liblego_hip_tune.so'))
from legommenders_amd import _lib
L = _lib.lib()
P, I = ctypes.c_void_p, ctypes.c_int
L.lego_debug_gemm_tn.argtypes = [I, I, P, P, P, I, I, I, P]
dev = torch.device('cuda:0')
def bench(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
names = {0: "128x128 8w", 1: "64x64 4w", 2: "128x64 4w", 3: "64x128 4w", 4: "128x128 4w", 20: "SPLIT 128x128 8w", 21: "SPLIT 64x64 4w", 22: "SPLIT 128x128 4w"}
tiles = {0: (128,128), 1: (64,64), 2: (128,64), 3: (64,128), 4:(128,128), 20: (128,128), 21: (64,64), 22: (128,128)}
VARIANTS = tuple(int(v) for v in sys.argv[1].split(',')) if len(sys.argv) > 1 else tuple(range(5))
SHAPES = [(26368, 256, 256), (26368, 256, 300), (27900, 256, 256), (1200, 256, 256)]
TARGETS = (256, 512, 1024)
if len(sys.argv) > 2 and sys.argv[2] == 'bert':      # BERT-base weight gradients at the rows of a B = 64 step
    SHAPES = [(29600, 768, 3072), (29600, 2304, 768), (29600, 768, 768)]
    TARGETS = (576, 1152, 2304, 4608)
for (R, N, K) in SHAPES:
    g = torch.randn(R, N, device=dev); x = torch.randn(R, K, device=dev)
    ref = g.double().T @ x.double()
    for v in VARIANTS:
        bm, bn = tiles[v]
        nt = ((N + bm - 1)//bm) * ((K + bn - 1)//bn)
        for target in TARGETS:
            split = max(1, min(target // nt, (R + 127)//128))
            dW = torch.zeros(N, K, device=dev)
            rc = L.lego_debug_gemm_tn(v, split, g.data_ptr(), x.data_ptr(), dW.data_ptr(), R, N, K, None)
            assert rc == 0, L.lego_last_error()
            torch.cuda.synchronize()
            err = (dW.double() - ref).abs().max().item() / ref.abs().max().item()
            ms = bench(lambda: L.lego_debug_gemm_tn(v, split, g.data_ptr(), x.data_ptr(), dW.data_ptr(), R, N, K, None))
            print(f"R={R} N={N} K={K} v{v} {names[v]:17s} split={split:4d} blocks={nt*split:5d}: {ms*1e3:7.1f} us {2*R*N*K/ms/1e9:6.1f} TF/s err {err:.0e}")
