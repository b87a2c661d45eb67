"""NRMS (GloVe) training trajectories with the in-projection per distinct key + sparse Dropout correction (LEGO_NRMS_DROPCORR=1, the
default) and row by row (=0), and -- as the yardstick -- the row form against itself with the attention block layer by layer
(fold level 0), same seeds and dropout streams: per-step losses.  The forms agree to rounding (forward q|k|v within 1.1e-6 of float64 in
both, tools/dropcorr_debug.py); Adam amplifies rounding differences, so the trajectories drift apart chaotically after a few steps --
what must NOT happen is a systematic offset, and the drift must look like the yardstick's.
    python tools/nrms_dropcorr_trajectory.py [steps]"""
import os
import statistics
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from legommenders_amd import engine as E  # noqa: E402
from legommenders_amd.synthetic import MIND_SMALL, glove_like, init_nrms_params, make_world  # noqa: E402
from legommenders_amd.train_step import DeviceData, TrainStep  # noqa: E402

dev = torch.device("cuda:0")
cfg = dict(MIND_SMALL)
world = make_world(seed=2023, **cfg)
glove = glove_like(cfg["V"], 300, seed=2024, device=dev)
N = int(sys.argv[1]) if len(sys.argv) > 1 else 600
out = {}
orig = E.NrmsEngine.__init__
for name, dc, level in (("rows", "0", 2), ("per_key", "1", 2), ("rows_unfolded", "0", 0)):
    os.environ["LEGO_NRMS_DROPCORR"] = dc

    def patched(self, *a, _lv=level, **kw):
        kw["fold_linear"] = _lv
        orig(self, *a, **kw)
    E.NrmsEngine.__init__ = patched
    data = DeviceData(world, dev, seed=2023)
    params = init_nrms_params(D=256, V=cfg["V"], n_cat=cfg["n_cat"], glove=glove)
    ts = TrainStep("nrms", params, data, 64, K=4, lr=1e-3, total_steps=0, seed=2023, dropout=True, tail="drop", glove=True)
    assert ts.engine.dropcorr == (dc == "1")
    losses = [ts.step().clone() for _ in range(N)]
    torch.cuda.synchronize()
    if dc == "1":
        assert ts.engine._dc_active, "the planned training step did not take the per-key in-projection"
    out[name] = [float(x) for x in losses]
    del ts
E.NrmsEngine.__init__ = orig
a, b, c = out["rows"], out["per_key"], out["rows_unfolded"]
for i in list(range(0, 10)) + list(range(10, N, max(1, N // 20))):
    print(f"step {i:4d}: rows {a[i]:.5f}  per key {b[i]:.5f} ({b[i]-a[i]:+.2e})  rows, unfolded {c[i]:.5f} ({c[i]-a[i]:+.2e})")
for lo in range(0, N, N // 4):
    hi = lo + N // 4
    print(f"mean loss steps {lo}-{hi}: rows {statistics.mean(a[lo:hi]):.4f}  per key {statistics.mean(b[lo:hi]):.4f}  rows, unfolded {statistics.mean(c[lo:hi]):.4f}")
rms = lambda x, y: (sum((p - q) ** 2 for p, q in zip(x, y)) / len(x)) ** 0.5
print(f"rms loss difference over steps {N // 2}-{N}: per key vs rows {rms(a[N // 2:], b[N // 2:]):.4f}, unfolded vs rows {rms(a[N // 2:], c[N // 2:]):.4f}")
