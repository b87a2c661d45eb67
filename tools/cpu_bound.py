import sys, os, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from legommenders_amd.synthetic import MIND_SMALL, glove_like, init_naml_params, init_nrms_params, make_world
from legommenders_amd.train_step import DeviceData, TrainStep
dev = torch.device('cuda:0')
cfg = dict(MIND_SMALL)
world = make_world(seed=2023, **cfg)
data = DeviceData(world, dev)
glove = glove_like(cfg["V"], 300, seed=2024, device=dev)
kind = sys.argv[1] if len(sys.argv) > 1 else "naml"          # naml | nrms | nrms_null
use_glove = kind != "nrms_null"
init = init_naml_params if kind == "naml" else init_nrms_params
params = init(D=256, V=cfg["V"], n_cat=cfg["n_cat"], glove=glove if use_glove else None)
ts = TrainStep(kind.split("_")[0], params, data, 64, glove=use_glove)
for _ in range(20): ts.step()
torch.cuda.synchronize()
for rep in range(3):
    t0 = time.perf_counter()
    for _ in range(200): ts.step()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f"enqueue {1e3*(t1-t0)/200:.3f} ms/step, total {1e3*(t2-t0)/200:.3f} ms/step")
import cProfile, pstats
pr = cProfile.Profile(); pr.enable()
for _ in range(100): ts.step()
pr.disable(); torch.cuda.synchronize()
pstats.Stats(pr).sort_stats('tottime').print_stats(45)
