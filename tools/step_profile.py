"""Per-step device time of the first N training steps (an event after every step, read at the end): how long the path takes
to reach its steady state after start-up (clock ramp, caches, allocator), and what the tagged-kernel event bracketing costs."""
import sys, os, torch
sys.path.insert(0, os.getcwd())
from legommenders_amd.synthetic import MIND_SMALL, glove_like, init_naml_params, make_world
from legommenders_amd.train_step import DeviceData, TrainStep
dev = torch.device("cuda:0")
cfg = dict(MIND_SMALL)
world = make_world(seed=2023, **cfg)
data = DeviceData(world, dev, seed=2023)
glove = glove_like(cfg["V"], 300, seed=2024, device=dev)
params = init_naml_params(D=256, V=cfg["V"], n_cat=cfg["n_cat"], glove=glove)
ts = TrainStep("naml", params, data, 64, K=4, lr=1e-3, total_steps=0, seed=2023, dropout=True, tail="drop", glove=True)
N = 120
if os.environ.get("NOGC") == "1":                   # ... or the host (Python's cyclic GC pausing the enqueue loop)?
    import gc
    gc.collect(); gc.disable()
if os.environ.get("SPIN_MS"):                       # busy the device first: is the slow start a clock ramp?
    a = torch.randn(4096, 4096, device=dev)
    t0 = torch.cuda.Event(enable_timing=True); t1 = torch.cuda.Event(enable_timing=True)
    t0.record()
    for _ in range(int(float(os.environ["SPIN_MS"]) / 1.0)):
        a = (a @ a).clamp_(-1, 1)
    t1.record(); torch.cuda.synchronize()
    print("spin", t0.elapsed_time(t1), "ms")
if os.environ.get("TOUCH") == "1":                  # ... or first touches of the 480 MB table?
    print("touch", float(glove.sum()))
evs = [torch.cuda.Event(enable_timing=True) for _ in range(N + 1)]
torch.cuda.synchronize()
evs[0].record()
timers = {}
rows = []
for i in range(N):
    ts.engine.timers = timers if (i >= 60 and i % 8 == 0) else None
    ts.step()
    rows.append(ts.engine.counters.clone())          # device copy of the batch's extents (token rows, instances, ..., distinct tokens)
    evs[i + 1].record()
torch.cuda.synchronize()
d = [evs[i].elapsed_time(evs[i + 1]) for i in range(N)]
for a in range(0, N, 10):
    print(f"steps {a:3d}-{a+9:3d}: " + " ".join(f"{x:.3f}" for x in d[a:a + 10]))
plain = [d[i] for i in range(60, N) if i % 8 != 0]
tagged = [d[i] for i in range(60, N) if i % 8 == 0]
print(f"steady plain {sum(plain)/len(plain):.4f} ms, steps with tagged-kernel events {sum(tagged)/len(tagged):.4f} ms")
print(f"mean of steps 5-24 (the driver's window): {sum(d[5:25])/20:.4f} ms; steps 40-59: {sum(d[40:60])/20:.4f}")
print("steps 0-29:", " ".join(f"{x:.3f}" for x in d[:30]))
rr = torch.stack(rows).cpu().numpy()
print("token rows per step, steps 0-29:", rr[:30, 0].tolist())
import numpy as np
print("corr(step time, token rows) over steps 30-119 (untagged):", float(np.corrcoef([d[i] for i in range(30, N) if i % 8 != 0], [rr[i, 0] for i in range(30, N) if i % 8 != 0])[0, 1]))
print("mean token rows steps 5-24: %.0f, steps 40-119: %.0f" % (rr[5:25, 0].mean(), rr[40:, 0].mean()))

