#!/usr/bin/env python3
"""Reference TRAINING band on a learnable world: tests/golden/train_band_{naml,nrms}.{json,npz}.

Runs ONLY in the build container (needs /root/reference).  It drives the REAL reference -- its `Legommender`, its
`Resampler` (negative sampling with python `random`), its `DataSet`, a `DataLoader(shuffle=True)` and the training loop of
`trainer.py:184-204` (Adam + HF linear schedule, `base_lego.py:201-223`), torch dropout ON (0.1 at the reference's three
sites) -- on `legommenders_amd.synthetic.make_learnable_world`, for several seeds, and scores the dev rows through the
reference's own `Legommender.forward` (test phase) + `MetricPool` (`base_lego.py:400-427`, `utils/metrics.py`).

Stored (data only): per seed GAUC / NDCG@10 / MRR before and after training, the mean training loss of the last 50 steps,
Both sides start every seed from `synthetic.init_*_params(seed)` (loaded into the reference model by its own state_dict
keys), so the MI355X run differs only in its random streams (Philox dropout, device sampler, its own shuffle).  `tests/test_train_band.py` (-m gpu) holds the HIP trainer's
mean GAUC to the reference's mean within 0.002 + the reference's seed spread.

    python tests/golden/make_train_band.py [--workers N] [naml|nrms|naml_d256|nrms_d256 ...]
"""
from __future__ import annotations

import json
import os
import random
import sys
import time

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)
import make_golden as MG                                              # noqa: E402  (stubs + duck-typed tables)
from legommenders_amd.synthetic import glove_table_np, init_naml_params, init_nrms_params, make_learnable_world   # noqa: E402

# Round 4: >= 16 seeds, a 4 000-user dev split (40 000 dev rows: the per-model evaluation noise of a
# 400-user split was most of NAML's seed spread), twice the training rows, NRMS trained to its plateau (10 epochs: at 5 it
# was mid-climb and the seeds spread over 0.055), and a second pair of bands at the HEADLINE width D = 256 / B = 64.
WORLD = dict(seed=0, n_items=1200, n_users=6000, n_rows=19200, V=3000, T=16, S=20, n_cat=18, neg_cap=20, p_pref=0.8,
             p_topic=0.5, pool=30, n_dev_users=4000, dev_neg=8)
HYPER = dict(D=64, B=32, lr=1e-3, epochs=2, dropout=0.1, heads=8, K=4, glove_seed=77)
# band name -> (model kind, overrides of HYPER, seeds[, overrides of WORLD])
BANDS = {
    "naml":      ("naml", dict(D=64, B=32, epochs=3), tuple(range(11, 27))),
    "nrms":      ("nrms", dict(D=64, B=32, epochs=10), tuple(range(11, 35))),      # 24 seeds: its seed std (0.003) is 3x NAML's
    "naml_d256": ("naml", dict(D=256, B=64, epochs=3), tuple(range(11, 27))),      # round 5: 16 seeds at the headline width too
    "nrms_d256": ("nrms", dict(D=256, B=64, epochs=5), tuple(range(11, 27))),
    # round 5: a MIND-shaped world -- titles of up to 30 tokens, histories of up to 50 clicks (the bench's T / S), same generator
    "naml_mind": ("naml", dict(D=64, B=32, epochs=2), tuple(range(11, 19)), dict(T=30, S=50)),
}


def band_world(band):
    return dict(WORLD, **(BANDS[band][3] if len(BANDS[band]) > 3 else {}))


METRICS = ["GAUC", "NDCG@10", "MRR"]
CACHE = os.environ.get("LEGO_BAND_CACHE", "/tmp/lego_band_cache")     # one json per (band, seed): the generator is resumable


def build(kind, w, seed, pretrained=True):
    """Manager.__init__ order (loader/manager.py:139-153,294-326) on duck-typed tables of the learnable world"""
    from loader.env import Env
    Env.device = torch.device("cpu")
    from loader.column_map import ColumnMap
    from loader.embedding_hub import EmbeddingHub
    from loader.resampler import Resampler
    from model.lego_config import LegoConfig
    from model.legommender import Legommender
    from model.operators.ada_operator import AdaOperator
    from model.operators.attention_operator import AttentionOperator
    from model.operators.cnn_operator import CNNOperator
    from model.predictors.dot_predictor import DotPredictor
    D, p, heads = HYPER["D"], HYPER["dropout"], HYPER["heads"]
    glove_v, cat_v = MG.SizedVocab("glove", w["V"]), MG.SizedVocab("category", w["n_cat"])
    item_v, user_v = MG.SizedVocab("item_id", w["n_items"]), MG.SizedVocab("user_id", w["n_users"])
    hist = [w["user_hist"][u, : w["user_hist_len"][u]].tolist() for u in range(w["n_users"])]
    neg = [w["neg_list"][u, : w["neg_len"][u]].tolist() for u in range(w["n_users"])]
    item_rows = [{"item_id": i, "title@glove": w["title_tok"][i, : w["title_len"][i]].tolist(), "category": int(w["cat"][i])}
                 for i in range(w["n_items"])]
    item_ut = MG.FakeUT(item_rows, [MG.Feat("item_id", item_v), MG.Feat("title@glove", glove_v, w["T"]),
                                    MG.Feat("category", cat_v)], "item_id")
    user_rows = [{"user_id": u, "history": list(hist[u]), "neg": list(neg[u])} for u in range(w["n_users"])]
    user_ut = MG.FakeUT(user_rows, [MG.Feat("user_id", user_v), MG.Feat("history", item_v, w["S"]),
                                    MG.Feat("neg", item_v, w["neg_cap"])], "user_id")

    def inter(users, items, labels):
        rows = [{"index": r, "user_id": int(users[r]), "item_id": int(items[r]), "click": int(labels[r]),
                 "history": list(hist[users[r]]), "neg": list(neg[users[r]])} for r in range(len(users))]
        return MG.FakeUT(rows, [MG.Feat("index", MG.SizedVocab("index", len(rows))), MG.Feat("user_id", user_v),
                                MG.Feat("item_id", item_v), MG.Feat("click", MG.SizedVocab("click", 2)),
                                MG.Feat("history", item_v, w["S"]), MG.Feat("neg", item_v, w["neg_cap"])], "index")
    train_ut = inter(w["row_user"], w["row_item"], np.ones(w["n_rows"], dtype=np.int64))
    v = w["valid"]
    dev_ut = inter(v["user"], v["item"], v["label"])
    if kind == "naml":
        lc = LegoConfig(hidden_size=D, item_hidden_size=D, neg_count=HYPER["K"],
                        user_config={"inputer_config": {"use_cls_token": False, "use_sep_token": False}},
                        item_config={"dropout": p, "kernel_size": 3})
        lc.set_component_classes(CNNOperator, AdaOperator, DotPredictor)
    else:
        lc = LegoConfig(hidden_size=D, item_hidden_size=D, neg_count=HYPER["K"],
                        item_config={"num_attention_heads": heads, "attention_dropout": p,
                                     "inputer_config": {"use_cls_token": False, "use_sep_token": True}},
                        user_config={"num_attention_heads": heads, "attention_dropout": p,
                                     "inputer_config": {"use_cls_token": False, "use_sep_token": False}})
        lc.set_component_classes(AttentionOperator, AttentionOperator, DotPredictor)
    lc.set_item_ut(item_ut, ["title@glove", "category"])
    lc.set_user_ut(user_ut, ["history"])
    lc.set_column_map(ColumnMap(item_col="item_id", user_col="user_id", history_col="history", neg_col="neg",
                                label_col="click", group_col="user_id"))
    eh = EmbeddingHub(embedding_dim=D, transformation="auto", transformation_dropout=p)
    if pretrained:                                # else (config/embed/null.yaml): a trainable nn.Embedding(V, D) token table
        path = "/tmp/_band_glove_%d_%d_%d.npy" % (HYPER["glove_seed"], w["V"], os.getpid())
        np.save(path, glove_table_np(HYPER["glove_seed"], w["V"]))
        eh.load_pretrained_embedding(path, vocab_name="glove", frozen=True)
        os.remove(path)
    eh.register_ut(item_ut, ["title@glove", "category"])
    lc.set_embedding_hub(eh)
    lc.build_components()
    lc.register_inputer_vocabs()
    torch.manual_seed(seed)                       # the model's initial parameters are a function of the seed alone
    model = Legommender(lc)
    return model, Resampler(lc), train_ut, dev_ut


def evaluate(model, resampler, dev_ut):
    """base_lego.py:349-427: test-phase scores of every dev row, then the reference's MetricPool"""
    from loader.data_set import DataSet
    from loader.env import Env
    from torch.utils.data import DataLoader
    from utils.metrics import MetricPool
    Env.test()
    model.eval()
    loader = DataLoader(DataSet(dev_ut, resampler), batch_size=200, shuffle=False)
    scores, labels, groups = [], [], []
    with torch.no_grad():
        for batch in loader:
            s = model(batch=batch)
            scores.append(s.squeeze(1) if s.dim() == 2 else s)
            labels.append(batch["click"])
            groups.append(batch["user_id"])
    scores, labels, groups = torch.cat(scores), torch.cat(labels), torch.cat(groups)
    res = MetricPool.parse(METRICS).calculate(scores.tolist(), labels.tolist(), groups.tolist())
    return {k: float(v) for k, v in res.items()}


def run_seed(kind, w, seed):
    from loader.data_set import DataSet
    from loader.env import Env
    from torch.utils.data import DataLoader
    from transformers import get_linear_schedule_with_warmup
    random.seed(seed); np.random.seed(seed)                           # utils/function.py:58-75 (seeding)
    model, resampler, train_ut, dev_ut = build(kind, w, seed)
    # initial parameters: OUR seeded initialiser (torch CPU generator: the same values on the GPU box), loaded into the reference
    # model by its own state_dict keys -- both sides then start from the same point without a parameter fixture
    A = int(model.state_dict()["item_op.additive_attention.encoder.0.weight"].shape[0])
    glove = torch.from_numpy(glove_table_np(HYPER["glove_seed"], w["V"]))
    if kind == "naml":
        P = init_naml_params(D=HYPER["D"], A=A, V=w["V"], n_cat=w["n_cat"], seed=seed, glove=glove)
    else:
        P = init_nrms_params(D=HYPER["D"], A=A, V=w["V"], n_cat=w["n_cat"], heads=HYPER["heads"], seed=seed, glove=glove)
    assert set(P) == set(model.state_dict()), set(P) ^ set(model.state_dict())
    for k, v in model.state_dict().items():
        assert tuple(v.shape) == tuple(P[k].shape), (k, v.shape, P[k].shape)
    model.load_state_dict(P)
    init = dict(A=A)
    before = evaluate(model, resampler, dev_ut)
    B, epochs = HYPER["B"], HYPER["epochs"]
    opt = torch.optim.Adam(filter(lambda p: p.requires_grad, model.parameters()), lr=HYPER["lr"])   # base_lego.py:201-204
    sched = get_linear_schedule_with_warmup(opt, num_warmup_steps=0, num_training_steps=len(train_ut) // B * epochs)
    torch.manual_seed(seed + 1000)                                    # shuffle + dropout streams
    loader = DataLoader(DataSet(train_ut, resampler), batch_size=B, shuffle=True)               # manager.py:374-381
    losses = []
    opt.zero_grad()
    for epoch in range(epochs):                                       # trainer.py:184-204
        model.train()
        Env.train()
        for batch in loader:
            loss = model(batch=batch)
            loss.backward()
            opt.step()
            sched.step()
            opt.zero_grad()
            losses.append(float(loss))
    after = evaluate(model, resampler, dev_ut)
    return init, dict(seed=seed, before=before, after=after, steps=len(losses), first_loss=float(np.mean(losses[:20])),
                      last_loss=float(np.mean(losses[-50:])))


def one(band, seed):
    """one (band, seed) run in THIS process (single thread: eight of them side by side use the cores better than one
    eight-thread run of these small products); result -> CACHE/<band>_<seed>.json"""
    kind, over = BANDS[band][:2]
    HYPER.update(over)
    MG.install_stubs()
    torch.set_num_threads(1)
    w = make_learnable_world(**band_world(band))
    t0 = time.time()
    init, r = run_seed(kind, w, seed)
    r["seconds"] = round(time.time() - t0, 1)
    os.makedirs(CACHE, exist_ok=True)
    json.dump(dict(init=init, run=r), open(os.path.join(CACHE, f"{band}_{seed}.json"), "w"))
    print(band, seed, "GAUC %.4f -> %.4f" % (r["before"]["GAUC"], r["after"]["GAUC"]), "NDCG@10 %.4f" % r["after"]["NDCG@10"],
          "loss %.4f -> %.4f" % (r["first_loss"], r["last_loss"]), "%.0f s" % r["seconds"], flush=True)


def assemble(band):
    kind, over, seeds = BANDS[band][:3]
    done = [json.load(open(os.path.join(CACHE, f"{band}_{s}.json"))) for s in seeds
            if os.path.exists(os.path.join(CACHE, f"{band}_{s}.json"))]
    if len(done) < len(seeds):
        print(band, "incomplete: %d of %d seeds" % (len(done), len(seeds)))
        return
    runs, inits = [d["run"] for d in done], {}
    for d in done:
        inits.update(d["init"])
    g = np.array([r["after"]["GAUC"] for r in runs])
    out = dict(kind=kind, band=band, world=band_world(band), hyper=dict(HYPER, **over, **inits), metrics=METRICS, runs=runs,
               init="legommenders_amd.synthetic.init_%s_params(D, A, V, n_cat, seed=run seed, glove=glove_table_np(glove_seed, V))" % kind,
               mean={m: float(np.mean([r["after"][m] for r in runs])) for m in METRICS},
               spread={m: float(np.max([r["after"][m] for r in runs]) - np.min([r["after"][m] for r in runs])) for m in METRICS},
               std={m: float(np.std([r["after"][m] for r in runs], ddof=1)) for m in METRICS},
               mean_before={m: float(np.mean([r["before"][m] for r in runs])) for m in METRICS},
               torch=torch.__version__,
               note="reference run: torch dropout, python-random negatives, DataLoader(shuffle=True), num_workers=0, 1 thread")
    json.dump(out, open(os.path.join(HERE, f"train_band_{band}.json"), "w"), indent=1)
    print(band, "mean GAUC %.4f std %.4f spread %.4f (%d seeds)" % (g.mean(), g.std(ddof=1), g.max() - g.min(), len(g)))


def main():
    """make_train_band.py [--workers N] [band ...]      run the missing (band, seed) jobs N at a time, then assemble
       make_train_band.py --one BAND SEED               (worker)
       make_train_band.py --assemble [band ...]"""
    import subprocess
    a = sys.argv[1:]
    if a[:1] == ["--one"]:
        return one(a[1], int(a[2]))
    workers = 5
    if a[:1] == ["--workers"]:
        workers, a = int(a[1]), a[2:]
    only_assemble = a[:1] == ["--assemble"]
    bands = [b for b in a if b in BANDS] or list(BANDS)
    if not only_assemble:
        jobs = [(b, s) for b in bands for s in BANDS[b][2] if not os.path.exists(os.path.join(CACHE, f"{b}_{s}.json"))]
        env = dict(os.environ, OMP_NUM_THREADS="1", MKL_NUM_THREADS="1")
        running = []
        while jobs or running:
            running = [p for p in running if p.poll() is None]
            while jobs and len(running) < workers:
                b, s = jobs.pop(0)
                running.append(subprocess.Popen([sys.executable, os.path.abspath(__file__), "--one", b, str(s)], env=env))
            time.sleep(2.0)
    for b in bands:
        assemble(b)


if __name__ == "__main__":
    main()
