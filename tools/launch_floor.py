import ctypes, os, sys, torch
sys.path.insert(0, os.getcwd())
from legommenders_amd._lib import call
dev = torch.device("cuda:0")
def P(t): return ctypes.c_void_p(t.data_ptr())
def bench(fn, n=200):
    for _ in range(20): fn()
    torch.cuda.synchronize()
    a, c = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    c.record(); torch.cuda.synchronize()
    return a.elapsed_time(c) / n * 1e3
x = torch.zeros(64, 256, device=dev)
cnt = torch.tensor([4], dtype=torch.int32, device=dev)
print("zero_rows tiny: %.2f us per launch" % bench(lambda: call("lego_zero_rows", P(x), 256, 256, 64, P(cnt), None)))
# strip kernel with almost no rows: launch + prologue + 8 tiles + epilogue of ONE workgroup
for R in (16, 112, 112 * 16, 112 * 64, 112 * 256):
    xx = torch.randn(max(R, 16), 256, device=dev); W = torch.randn(256, 256, device=dev) * 0.05; b = torch.randn(256, device=dev)
    t = torch.zeros(max(R, 16), 256, device=dev); c = torch.tensor([R], dtype=torch.int32, device=dev)
    print("NT tanh rows %6d: %.2f us" % (R, bench(lambda: call("lego_linear_fwd", P(xx), 256, P(W), 256, P(b), P(t), 256, R, P(c), 256, 256, 2, None, None, None, None, None), 100)))
