import sys, numpy as np, torch
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests.golden_util import load_model_fixture
from legommenders_amd import engine as E
name = sys.argv[1]
dev = torch.device('cuda:0')
meta, P, G, tables, batch, logits, loss = load_model_fixture(name)
Pd = {k: torch.tensor(v).to(dev).contiguous() for k, v in P.items()}
tb = E.ItemTables(tables["title_tok"], tables["title_len"], tables["cat"], dev)
B, C = batch["cand"].shape; S = batch["hist"].shape[1]
if meta["kind"] == "naml":
    eng = E.NamlEngine(Pd, tb, B, C, S)
else:
    eng = E.NrmsEngine(Pd, tb, B, C, S, heads=meta["heads"], glove=(meta["embed"] == "glove"))
ids = [torch.tensor(batch[k]).int().to(dev).contiguous() for k in ("cand", "hist", "hist_len")]
scores, l = eng.forward(*ids, training=False)
print("logits maxdiff", np.abs(scores.cpu().numpy() - logits).max(), "loss", float(l), loss)
grads = eng.grads_like(); eng.backward(grads); torch.cuda.synchronize()
for k, g in G.items():
    got = grads[k].cpu().numpy(); d = np.abs(got - g); sc = np.abs(g).max()
    idx = np.unravel_index(np.argmax(d), d.shape)
    print(f"{k:60s} max {d.max():.2e} rel {d.max()/max(sc,1e-30):.2e} n>1e-4rel {(d > 1e-4*sc).sum()} at {idx} got {got[idx]:.6e} ref {g[idx]:.6e}")
