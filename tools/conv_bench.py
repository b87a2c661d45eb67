import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
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
D = 256
for (n, L) in [(1551, 17), (879, 30), (1024, 32), (2048, 16)]:
    lens = torch.full((n,), L)
    mask = torch.ones(n, L, dtype=torch.int32, device=dev)
    R = n * L
    h = torch.randn(R, D, device=dev); w = torch.randn(D, D, 3, device=dev) * 0.05; b = torch.randn(D, device=dev)
    plan = K.plan_dense(mask); wt = K.conv3_pack(w)
    for drop in (None, (0.1, 1, 2)):
        ms = bench(lambda: K.conv3_fwd(h, wt, b, plan, drop=drop))
        print(f"conv3_fwd R={R} drop={drop is not None}: {ms*1e3:7.1f} us {2*R*D*3*D/ms/1e9:6.1f} TF/s")
    x = torch.randn(R, 3 * D, device=dev); W = torch.randn(D, 3 * D, device=dev) * 0.05
    y = torch.empty(R, D, device=dev)
    ms = bench(lambda: K.linear_fwd(x, W, b, act=0, out=y))
    print(f"plain NT   R={R}            : {ms*1e3:7.1f} us {2*R*D*3*D/ms/1e9:6.1f} TF/s")
    ms = bench(lambda: K.linear_fwd(x, W, b, act=1, out=y, rowinfo=plan.rowinfo, drop=(0.1, 1, 2)))
    print(f"NT+relu+live+drop R={R}     : {ms*1e3:7.1f} us {2*R*D*3*D/ms/1e9:6.1f} TF/s")
