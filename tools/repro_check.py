"""Run-to-run reproducibility of the training step: the same seeded training repeated in one process; per-step losses
should agree to atomics-order noise (~1e-5).  A larger jump at some step points at a race."""
import os, sys, torch, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from legommenders_amd.synthetic import MIND_SMALL, make_world, glove_like, init_naml_params
from legommenders_amd.train_step import DeviceData, TrainStep
dev = torch.device("cuda:0")
cfg = dict(MIND_SMALL)
world = make_world(seed=2023, **cfg)
data = DeviceData(world, dev, seed=2023)
glove = glove_like(cfg["V"], 300, seed=2024, device=dev)
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 120
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 6
ref = None
for r in range(reps):
    params = init_naml_params(D=256, V=cfg["V"], n_cat=cfg["n_cat"], glove=glove)
    ts = TrainStep("naml", params, data, 64, K=4, lr=1e-3, total_steps=0, seed=2023, dropout=True)
    losses = []
    for i in range(steps):
        losses.append(ts.step().clone())
    torch.cuda.synchronize()
    l = torch.stack(losses).flatten().cpu().numpy()
    if ref is None:
        ref = l
        print("run 0 final", l[-1])
        continue
    d = np.abs(l - ref)
    first = int(np.argmax(d > 2e-4)) if (d > 2e-4).any() else -1
    print(f"run {r}: max|dloss| {d.max():.2e} at step {int(d.argmax())}; first step with |d| > 2e-4: {first}; final {l[-1]:.5f}")
