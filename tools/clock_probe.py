"""In-kernel clock of the two big GEMM kernels during training steps (tuning build: `make -C legommenders_amd/csrc tune`).
clock = delta(s_memtime) / delta(s_memrealtime) x 100 MHz, summed over workgroups and launches (MI355X_MICROARCH.md, DVFS give-back 6)."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("LEGO_HIP_LIB", os.path.join(ROOT, "legommenders_amd", "csrc", "liblego_hip_tune.so"))
import torch
from legommenders_amd import _lib
from legommenders_amd.synthetic import MIND_SMALL, glove_like, init_naml_params, make_world
from legommenders_amd.train_step import DeviceData, TrainStep
dev = torch.device("cuda:0")
w = make_world(seed=2023, **MIND_SMALL)
P = init_naml_params(D=256, V=w["V"], n_cat=w["n_cat"], glove=glove_like(w["V"], 300, seed=2024, device=dev))
ts = TrainStep("naml", P, DeviceData(w, dev, seed=2023), 64, tail="drop")
L = _lib.lib()
buf = (ctypes.c_ulonglong * 4)()
for _ in range(3000):          # > 2 s of back-to-back steps before reading (the clock settles)
    ts.step()
torch.cuda.synchronize()
L.lego_debug_clock(buf, 1)
for _ in range(500):
    ts.step()
torch.cuda.synchronize()
L.lego_debug_clock(buf, 0)
for name, i in (("winograd conv kernel", 0), ("row-strip kernel", 1)):
    c, r = buf[2 * i], buf[2 * i + 1]
    print(f"{name}: {c / max(r, 1) * 0.1:.3f} GHz in-kernel ({c} core cycles / {r} ticks of 10 ns)")
