"""Cost of HIP events between back-to-back kernels: torch events vs raw hipEvents with hipEventDisableSystemFence."""
import sys, os, time, ctypes, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from legommenders_amd import _lib
_lib.lib()
hip = ctypes.CDLL(os.path.join(os.path.dirname(torch.__file__), "lib", "libamdhip64.so"), mode=ctypes.RTLD_GLOBAL)
hip.hipEventCreateWithFlags.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_uint]
hip.hipEventRecord.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
hip.hipStreamWaitEvent.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint]
DISABLE_TIMING, DISABLE_SYSFENCE = 0x2, 0x20000000
def mk(flags):
    e = ctypes.c_void_p()
    rc = hip.hipEventCreateWithFlags(ctypes.byref(e), flags)
    assert rc == 0, rc
    return e
dev = torch.device("cuda:0")
x = torch.zeros(1 << 22, device=dev); y = torch.zeros(1 << 22, device=dev)
side = torch.cuda.Stream(dev); m = torch.cuda.current_stream()
ms, ss = ctypes.c_void_p(m.cuda_stream), ctypes.c_void_p(side.cuda_stream)
def chain(n, mode, flags=None):
    if flags is not None:
        evs = [mk(flags) for _ in range(n)]; evs2 = [mk(flags) for _ in range(n)]
    else:
        evs = [torch.cuda.Event() for _ in range(n)]; evs2 = [torch.cuda.Event() for _ in range(n)]
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(n):
        x.add_(1.0)
        if mode == "record":
            if flags is None: evs[i].record(m)
            else: hip.hipEventRecord(evs[i], ms)
        elif mode == "fork-join":
            if flags is None:
                evs[i].record(m); side.wait_event(evs[i])
            else:
                hip.hipEventRecord(evs[i], ms); hip.hipStreamWaitEvent(ss, evs[i], 0)
            with torch.cuda.stream(side):
                y.add_(1.0)
            if flags is None:
                evs2[i].record(side); m.wait_event(evs2[i])
            else:
                hip.hipEventRecord(evs2[i], ss); hip.hipStreamWaitEvent(ms, evs2[i], 0)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e6
for name, flags in (("torch", None), ("raw notiming", DISABLE_TIMING), ("raw notiming+nosysfence", DISABLE_TIMING | DISABLE_SYSFENCE)):
    for mode in ("plain", "record", "fork-join"):
        chain(100, mode, flags)
        print(f"{name:26s} {mode:10s} {chain(1000, mode, flags):7.2f} us per link")
