"""Is the main stream idle between the forward's last kernel and the backward's first one?  HIP events at the end of
engine.forward and at the start of engine.backward of every step (GPU time between them = idle or prefetch-only time when the
host enqueues late; ~0 when the host runs ahead).   python tools/neck_probe.py [naml|nrms]"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from legommenders_amd.synthetic import MIND_SMALL, glove_like, init_naml_params, init_nrms_params, make_world
from legommenders_amd.train_step import DeviceData, TrainStep
kind = sys.argv[1] if len(sys.argv) > 1 else "naml"
dev = torch.device("cuda:0")
cfg = dict(MIND_SMALL)
world = make_world(seed=2023, **cfg)
data = DeviceData(world, dev, seed=2023)
glove = glove_like(cfg["V"], 300, seed=2024, device=dev)
init = init_naml_params if kind == "naml" else init_nrms_params
ts = TrainStep(kind, init(D=256, V=cfg["V"], n_cat=cfg["n_cat"], glove=glove), data, 64, tail="drop")
eng = ts.engine
ev = {"f0": [], "f1": [], "b0": [], "b1": []}
f, b = eng.forward, eng.backward


def fwd(*a, **k):
    e0 = torch.cuda.Event(enable_timing=True); e0.record(); ev["f0"].append(e0)
    r = f(*a, **k)
    e1 = torch.cuda.Event(enable_timing=True); e1.record(); ev["f1"].append(e1)
    return r


def bwd(*a, **k):
    e0 = torch.cuda.Event(enable_timing=True); e0.record(); ev["b0"].append(e0)
    r = b(*a, **k)
    e1 = torch.cuda.Event(enable_timing=True); e1.record(); ev["b1"].append(e1)
    return r


import time
for _ in range(300):
    ts.step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(200):
    ts.step()
torch.cuda.synchronize()
print(kind, "plain: %.1f us per step" % ((time.perf_counter() - t0) / 200 * 1e6))
eng.forward, eng.backward = fwd, bwd
t0 = time.perf_counter()
for _ in range(200):
    ts.step()
torch.cuda.synchronize()
print(kind, "with 4 events per step: %.1f us per step" % ((time.perf_counter() - t0) / 200 * 1e6))
n = len(ev["f0"])
m = lambda x, y: sum(a.elapsed_time(b) for a, b in zip(ev[x][20:], ev[y][20:])) / (n - 20) * 1e3
print(kind, "forward %.1f us, forward end -> backward start %.1f us, backward %.1f us, backward end -> next forward start %.1f us" %
      (m("f0", "f1"), m("f1", "b0"), m("b0", "b1"), sum(a.elapsed_time(b) for a, b in zip(ev["b1"][20:-1], ev["f0"][21:])) / (n - 21) * 1e3))
