"""BERT-base block products (rows ~ 30 k live tokens of a B = 64 step) on the path's fp32 MFMA kernels against PyTorch-ROCm's
hipBLASLt path, per product form -- decides which library each nn.Linear of the BERT news encoder uses in each direction
(config 5).   python tools/bert_shapes_bench.py [rows]"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from legommenders_amd import kernels as K
dev = torch.device("cuda:0")
R = int(sys.argv[1]) if len(sys.argv) > 1 else 29600
torch.backends.cuda.matmul.allow_tf32 = False


def t(fn, reps=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3


for name, N, Kd in (("qkv/out 768x768", 768, 768), ("qkv fused 2304x768", 2304, 768), ("ffn1 3072x768", 3072, 768), ("ffn2 768x3072", 768, 3072)):
    x = torch.randn(R, Kd, device=dev)
    W = torch.randn(N, Kd, device=dev) * 0.02
    b = torch.randn(N, device=dev)
    g = torch.randn(R, N, device=dev)
    gW = torch.zeros(N, Kd, device=dev)
    fl = 2.0 * R * N * Kd
    y = torch.empty(R, N, device=dev)
    rows = []
    rows.append(("fwd  NT", t(lambda: torch.addmm(b, x, W.t(), out=y)), t(lambda: K.linear_fwd(x, W, b, out=y))))
    gx = torch.empty(R, Kd, device=dev)
    rows.append(("dx   NN", t(lambda: torch.mm(g, W, out=gx)), t(lambda: K.linear_bwd_data(g, W))))
    rows.append(("dW   TN", t(lambda: torch.mm(g.t(), x, out=gW)), t(lambda: K.linear_bwd_weight(g, x, gW))))
    for what, tt, tl in rows:
        print(f"{name:20s} {what}  torch {tt:8.1f} us {fl / tt / 1e6:6.1f} TF   lego {tl:8.1f} us {fl / tl / 1e6:6.1f} TF")
    # correctness of ours vs torch on this shape
    K.linear_fwd(x, W, b, out=y)
    ref = torch.addmm(b, x, W.t())
    print(f"{'':20s} max|lego - torch| fwd {float((y - ref).abs().max()):.2e} (scale {float(ref.abs().max()):.2e})")
