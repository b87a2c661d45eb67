"""Sweep of the HBM-bound row gather's launch shape (rows in flight per wave, streaming policy, grid) and of HOW the caches are made cold
before each launch: a 640 MB WRITE leaves 256 MB of dirty lines in the Infinity Cache that the gather's own traffic then evicts (their
write-back competes with the gather for HBM), a 640 MB READ leaves clean lines.   python tools/gather_sweep.py"""
import ctypes, os, subprocess, sys, json
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

if len(sys.argv) > 1 and sys.argv[1] == "one":
    from legommenders_amd._lib import call
    dev = torch.device("cuda:0")
    rows, V, E0 = 105600, 400000, 300
    g = torch.Generator(device="cpu").manual_seed(5)
    table = torch.randn(V, E0, device=dev)
    idx = torch.randint(0, V, (rows,), generator=g, dtype=torch.int32).to(dev)
    out = torch.empty(rows, E0, device=dev)
    cnt = torch.tensor([rows], dtype=torch.int32, device=dev)
    flush = torch.empty(160 << 20, dtype=torch.float32, device=dev)
    P = lambda t: ctypes.c_void_p(t.data_ptr())
    f = lambda: call("lego_gather_rows", P(table), E0, E0, P(idx), rows, P(cnt), P(out), E0, 0, None)
    res = {}
    for kind in ("write", "read"):
        f()
        ts = []
        for _ in range(10):
            if kind == "write":
                flush.add_(1.0)
            else:
                flush.sum()
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record(); f(); b.record()
            torch.cuda.synchronize()
            ts.append(a.elapsed_time(b))
        ms = sorted(ts)[len(ts) // 2]
        res[kind] = (round(ms * 1e3, 1), round(rows * (E0 * 8 + 4) / (ms * 1e-3) / 1e12, 3))
    print(json.dumps(res))
    sys.exit(0)

for u in ("4", "8", "2"):
    for nt in ("2", "0"):
        for blocks in ("1024", "2048", "4096"):
            env = dict(os.environ, LEGO_GATHER_U=u, LEGO_GATHER_NT=nt, LEGO_GATHER_BLOCKS=blocks)
            r = subprocess.run([sys.executable, os.path.abspath(__file__), "one"], env=env, capture_output=True, text=True)
            line = [l for l in r.stdout.splitlines() if l.startswith("{")]
            print(f"U={u} NT={nt} blocks={blocks}: (us, TB/s read+write) {line[0] if line else r.stderr[-300:]}", flush=True)
