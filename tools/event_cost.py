"""GPU-side cost of HIP event record / wait between back-to-back kernels on one stream (rocprof-free: wall time of a
long chain divided by its length)."""
import sys, os, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
dev = torch.device("cuda:0")
x = torch.zeros(1 << 20, device=dev)
side = torch.cuda.Stream(dev)
m = torch.cuda.current_stream()

def chain(n, mode):
    evs = [torch.cuda.Event() for _ in range(n)]
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(n):
        x.add_(1.0)
        if mode == "record":
            evs[i].record(m)
        elif mode == "record+sidewait":
            evs[i].record(m); side.wait_event(evs[i])
        elif mode == "fork-join":
            evs[i].record(m); side.wait_event(evs[i])
            with torch.cuda.stream(side):
                y.add_(1.0)
            e2 = torch.cuda.Event(); e2.record(side); m.wait_event(e2)
        elif mode == "wait-done":
            m.wait_event(done)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e6

y = torch.zeros(1 << 20, device=dev)
done = torch.cuda.Event(); done.record(side); torch.cuda.synchronize()
for mode in ("plain", "record", "record+sidewait", "wait-done", "fork-join"):
    chain(200, mode)
    print(f"{mode:18s} {chain(2000, mode):7.2f} us per link")
