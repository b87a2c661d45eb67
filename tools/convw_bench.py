"""Weight gradient of the k=3 conv at the bench shape: the direct 3-tap TN product (lego_conv3_bwd_weight) beside plain TN
products of the same size and the Winograd-domain form the engine uses."""
import sys, os, torch
sys.path.insert(0, os.getcwd())
from legommenders_amd import kernels as K
dev = torch.device("cuda:0")
def bench(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3
D = 256
for (n, L) in [(1551, 17), (879, 30)]:
    mask = torch.ones(n, L, dtype=torch.int32, device=dev)
    R = n * L
    gy = torch.randn(R, D, device=dev); h = torch.randn(R, D, device=dev)
    plan = K.plan_dense(mask)
    dwt = torch.zeros(3, D, D, device=dev)
    t = bench(lambda: K.conv3_bwd_weight(gy, h, plan, dwt)); fl = 2.0 * R * D * 3 * D
    print(f"R={R}: direct 3-tap TN        {t:7.1f} us  {fl/t/1e6:6.1f} TF/s")
    dW = torch.zeros(D, D, device=dev)
    t1 = bench(lambda: K.linear_bwd_weight(gy, h, dW))
    print(f"R={R}: one plain TN 256x256   {t1:7.1f} us  {fl/3/t1/1e6:6.1f} TF/s   (x3 = {3*t1:.1f} us)")
    h3 = torch.randn(R, 3 * D, device=dev); dW3 = torch.zeros(D, 3 * D, device=dev)
    t3 = bench(lambda: K.linear_bwd_weight(gy, h3, dW3))
    print(f"R={R}: plain TN 256x768       {t3:7.1f} us  {fl/t3/1e6:6.1f} TF/s")
