"""Does a captured HIP graph shed the cross-stream hand-off cost?  The same chain of (main kernel; fork -> side kernel
-> join) links, eager vs captured into one torch.cuda.CUDAGraph and replayed."""
import time, torch
dev = torch.device("cuda:0")
x = torch.zeros(1 << 22, device=dev); y = torch.zeros(1 << 22, device=dev)
side = torch.cuda.Stream(dev)
N = 40

def chain(fork):
    m = torch.cuda.current_stream()
    for i in range(N):
        x.add_(1.0)
        if fork:
            e = torch.cuda.Event(); e.record(m); side.wait_event(e)
            with torch.cuda.stream(side):
                y.add_(1.0)
            e2 = torch.cuda.Event(); e2.record(side); m.wait_event(e2)

def timeit(fn, reps=50):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps / N * 1e6

for fork in (False, True):
    eager = timeit(lambda: chain(fork))
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream(dev)
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        chain(fork)                      # warm-up on the capture stream
    torch.cuda.current_stream().wait_stream(s)
    torch.cuda.synchronize()
    with torch.cuda.graph(g):
        chain(fork)
    graph = timeit(g.replay)
    print(f"fork={fork}: eager {eager:6.2f} us per link, graph replay {graph:6.2f} us per link")
