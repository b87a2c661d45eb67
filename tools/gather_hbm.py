"""The row gather (lego_gather_rows -> gather_rows_wave_kernel) where it IS bound by HBM: 105 600 random 1 200-byte rows of the
480 MB GloVe-shaped table (the dense reference layout of one NAML batch: 64 x 55 items x 30 tokens), every launch on a cold
Infinity Cache (640 MB written in between).  Prints one JSON line; `measure()` is what bench.py's secondary section calls.
    python tools/gather_hbm.py [rows] [launches]        (under rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE for the traffic)"""
import ctypes
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

PEAK_HBM_GBS = 8000.0


def measure(dev, table=None, rows=105600, launches=8, V=400000, E0=300, zipf=False, seed=5, flush_kind="write"):
    """`flush_kind`: how the caches are made cold before every launch -- "write": 640 MB written (the last 256 MB stay in the Infinity Cache as
    DIRTY lines that the gather's own traffic has to evict: their write-back competes with it), "read": 640 MB read (cold, clean lines)"""
    from legommenders_amd._lib import call

    def P(t):
        return ctypes.c_void_p(t.data_ptr())
    g = torch.Generator(device="cpu").manual_seed(seed)
    if table is None:
        table = torch.randn(V, E0, device=dev)
    V = table.shape[0]
    if zipf:       # the bench world's token law (most rows repeat: the Infinity Cache serves them)
        import numpy as np
        idx = torch.from_numpy(np.minimum(np.random.RandomState(seed).zipf(1.2, size=rows) - 1, V - 1).astype("int32")).to(dev)
    else:
        idx = torch.randint(0, V, (rows,), generator=g, dtype=torch.int32).to(dev)
    out = torch.empty(rows, E0, device=dev)
    cnt = torch.tensor([rows], dtype=torch.int32, device=dev)
    flush = torch.empty(160 << 20, dtype=torch.float32, device=dev)          # 640 MB: 2.5 x the Infinity Cache
    f = lambda: call("lego_gather_rows", P(table), E0, E0, P(idx), rows, P(cnt), P(out), E0, 0, None)
    f()
    ts = []
    for _ in range(launches):
        if flush_kind == "write":
            flush.add_(1.0)
        else:
            flush.sum()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); f(); b.record()
        torch.cuda.synchronize()
        ts.append(a.elapsed_time(b))
    assert torch.equal(out[:64], table[idx[:64].long()])
    ms = sorted(ts)[len(ts) // 2]
    nbytes = rows * (E0 * 4 * 2 + 4)
    gbs = nbytes / (ms * 1e-3) / 1e9
    return {"kernel": "gather_rows_wave_kernel (lego_gather_rows)", "bound": "hbm", "rows": rows, "row_bytes": E0 * 4, "table_MB": round(V * E0 * 4 / 1e6, 1),
            "index_law": "zipf(1.2)" if zipf else "uniform", "algorithmic_bytes_per_launch": nbytes, "avg_launch_ms": round(ms, 5),
            "launch_ms_all": [round(t, 5) for t in ts], "achieved": round(gbs, 1), "peak": PEAK_HBM_GBS, "unit": "GB/s", "frac": round(gbs / PEAK_HBM_GBS, 4),
            "reads_only_GBs": round(rows * E0 * 4 / (ms * 1e-3) / 1e9, 1),
            "timed": f"median of {launches} launches, each behind a 640 MB {flush_kind} (cold Infinity Cache" + (", dirty lines" if flush_kind == "write" else ", clean lines") + "), HIP events around the launch"}


if __name__ == "__main__":
    pos = [a for a in sys.argv[1:] if not a.startswith("--")]
    rows = int(pos[0]) if len(pos) > 0 else 105600
    n = int(pos[1]) if len(pos) > 1 else 8
    d = torch.device("cuda:0")
    res = {"uniform": measure(d, rows=rows, launches=n), "uniform_clean_cache": measure(d, rows=rows, launches=n, flush_kind="read")}
    if "--uniform-only" not in sys.argv:            # (the PMC passes: every row-gather launch of the process is then an HBM-sized one)
        res["zipf"] = measure(d, rows=rows, launches=n, zipf=True)
    print(json.dumps(res))
