"""NRMS training trajectories with the attention block folded (level 2, fused user head, prefetched decode / gather) and layer by layer
(level 0), same seeds and dropout streams: per-step losses over a few hundred steps.  They agree to rounding for the first steps and
drift apart chaotically afterwards (Adam amplifies rounding differences); what must NOT happen is a systematic offset."""
import sys, os, torch
sys.path.insert(0, os.getcwd())
from legommenders_amd import engine as E
from legommenders_amd.synthetic import MIND_SMALL, glove_like, init_nrms_params, make_world
from legommenders_amd.train_step import DeviceData, TrainStep
dev = torch.device("cuda:0")
cfg = dict(MIND_SMALL)
world = make_world(seed=2023, **cfg)
glove = glove_like(cfg["V"], 300, seed=2024, device=dev)
N = int(sys.argv[1]) if len(sys.argv) > 1 else 600
out = {}
orig = E.NrmsEngine.__init__
for level in (2, 0):
    def patched(self, *a, _lv=level, **kw):
        kw["fold_linear"] = _lv
        orig(self, *a, **kw)
    E.NrmsEngine.__init__ = patched
    data = DeviceData(world, dev, seed=2023)
    params = init_nrms_params(D=256, V=cfg["V"], n_cat=cfg["n_cat"], glove=glove)
    ts = TrainStep("nrms", params, data, 64, K=4, lr=1e-3, total_steps=0, seed=2023, dropout=True, tail="drop", glove=True)
    losses = []
    for i in range(N):
        losses.append(ts.step().clone())
    torch.cuda.synchronize()
    out[level] = [float(x) for x in losses]
    del ts
E.NrmsEngine.__init__ = orig
a, b = out[2], out[0]
for i in list(range(0, 10)) + list(range(10, N, max(1, N // 20))):
    print(f"step {i:4d}: folded {a[i]:.5f}  layer-by-layer {b[i]:.5f}  diff {a[i]-b[i]:+.2e}")
import statistics
for lo in range(0, N, N // 4):
    hi = lo + N // 4
    print(f"mean loss steps {lo}-{hi}: folded {statistics.mean(a[lo:hi]):.4f}  layer-by-layer {statistics.mean(b[lo:hi]):.4f}")
