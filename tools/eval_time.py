"""Wall time of the evaluation path on the full MIND-small-shaped synthetic world: representation caches + scoring + metrics."""
import sys, time, torch, numpy as np
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from legommenders_amd.synthetic import MIND_SMALL, make_world, glove_like, init_naml_params
from legommenders_amd.train_step import DeviceData
from legommenders_amd.evaluate import Evaluator
dev = torch.device("cuda:0")
cfg = dict(MIND_SMALL); w = make_world(seed=2023, **cfg)
data = DeviceData(w, dev)
P = {k: v.to(dev).contiguous() for k, v in init_naml_params(D=256, V=cfg["V"], n_cat=cfg["n_cat"], glove=glove_like(cfg["V"], 300, seed=2024, device=dev)).items()}
ev = Evaluator("naml", P, data)
for _ in range(2):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    ev.build_caches(); torch.cuda.synchronize()
    print("caches: %.3f s for %d items + %d users" % (time.perf_counter() - t0, data.n_items, data.user_hist.shape[0]))
rs = np.random.RandomState(0)
u = np.repeat(rs.choice(data.user_hist.shape[0], size=20000, replace=False), 10); n = u.size      # 10 rows per user, one positive
it = rs.randint(0, data.n_items, n); lab = np.zeros(n, dtype=np.int64); lab[::10] = 1
t0 = time.perf_counter(); res, _ = ev.evaluate(u, it, lab); print("evaluate 200k rows: %.3f s" % (time.perf_counter() - t0), res)
from legommenders_amd import metrics as M
names = ["GAUC", "MRR", "NDCG@1", "NDCG@5", "NDCG@10"]
s = ev.scores(torch.as_tensor(u), torch.as_tensor(it)); torch.cuda.synchronize()
for _ in range(2):
    t0 = time.perf_counter(); r = M.calculate_device(s, lab, u, names); torch.cuda.synchronize()
    print("grouped metrics on the device (sort + launch + table copy): %.4f s" % (time.perf_counter() - t0))
t0 = time.perf_counter(); h = M.calculate(s.cpu().numpy(), lab, u, names)
print("same metrics, host form: %.3f s; max |diff| %.2e" % (time.perf_counter() - t0, max(abs(r[k] - h[k]) for k in names)))
