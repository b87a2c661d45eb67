#!/usr/bin/env python3
"""Generate the golden parity fixtures under tests/golden/ from the REAL reference.

Runs ONLY in the build container (needs /root/reference, read-only).  It imports the
reference's hot-path modules with two in-memory stubs for the un-vendored packages
(`unitok`, `pigmento`; recipe: SURVEY.md Appendix B), drives `Legommender.forward` /
the individual operators on seeded synthetic MIND-shaped inputs and stores *data only*
(inputs, parameters, expected outputs, expected gradients) as .npz files.  Nothing of
the reference's source text is copied; the fixtures are the pins the oracle
(`oracle/lego_oracle.py`) and the HIP kernels are tested against.

    python tests/golden/make_golden.py            # rewrites tests/golden/*.npz
"""
from __future__ import annotations

import json
import os
import random
import sys
import types

import numpy as np
import torch

REF = "/root/reference"
OUT = os.path.dirname(os.path.abspath(__file__))


# --------------------------------------------------------------------------- stubs
def install_stubs():
    sys.dont_write_bytecode = True
    if REF not in sys.path:
        sys.path.insert(0, REF)
    unitok = types.ModuleType("unitok")

    class Vocab:
        def __init__(self, name):
            self.name = name
            self._toks = []

        def append(self, tok):
            self._toks.append(tok)
            return len(self._toks) - 1

        @property
        def size(self):
            return len(self._toks)

    class Symbol:
        def __init__(self, name):
            self.name = name

    unitok.Vocab = Vocab
    unitok.Symbol = Symbol
    unitok.UniTok = type("UniTok", (), {})
    unitok.Feature = type("Feature", (), {})
    sys.modules["unitok"] = unitok
    pig = types.ModuleType("pigmento")
    pig.pnt = lambda *a, **k: None
    sys.modules["pigmento"] = pig


class SizedVocab:
    def __init__(self, name, size):
        self.name, self.size = name, size


class Feat:
    def __init__(self, name, vocab, max_len=None):
        self.name, self.max_len = name, max_len
        self.tokenizer = types.SimpleNamespace(vocab=vocab)


class FakeUT:
    def __init__(self, rows, feats, key):
        self.rows = rows
        self.meta = types.SimpleNamespace(features={f.name: f for f in feats})
        self.key_feature = key

    def __len__(self):
        return len(self.rows)

    def __getitem__(self, i):
        return dict(self.rows[i])


# --------------------------------------------------------------------------- synthetic MIND-shaped world
def glove_table(seed, V, E0=300):
    """Deterministic stand-in for data/embeddings/glove.npy (legacy RandomState is frozen)."""
    return (np.random.RandomState(seed).standard_normal((V, E0)) * 0.4).astype(np.float32)


def make_world(seed, V, n_items, n_users, n_rows, T=30, S=50, n_cat=18):
    rs = np.random.RandomState(seed)
    title_len = rs.randint(5, T + 1, size=n_items)
    title_len[0] = T          # item 0 is the history pad item: make it a full-length one
    title_len[1] = 1          # shortest legal title
    title_tok = -np.ones((n_items, T), dtype=np.int64)
    for i in range(n_items):
        # Zipf-ish token ids, id 0 is legal (and is what the pad rewrites to)
        z = rs.zipf(1.3, size=title_len[i]) - 1
        title_tok[i, : title_len[i]] = np.minimum(z, V - 1)
    cat = rs.randint(0, n_cat, size=n_items).astype(np.int64)
    hist_len = np.clip(rs.geometric(1 / 12.0, size=n_users), 1, S)
    hist_len[0] = S           # a full history
    hist_len[1] = 1
    hist = [rs.randint(0, n_items, size=hist_len[u]).tolist() for u in range(n_users)]
    neg_len = rs.randint(0, 9, size=n_users)
    neg_len[0] = 0            # no true negatives -> 4 random fills
    neg_len[1] = 2            # 2 true + 2 random
    neg = [rs.randint(0, n_items, size=neg_len[u]).tolist() for u in range(n_users)]
    users = rs.randint(0, n_users, size=n_rows)
    users[:2] = [0, 1]
    pos_items = rs.randint(0, n_items, size=n_rows)
    return dict(title_tok=title_tok, title_len=title_len.astype(np.int64), cat=cat,
                hist=hist, hist_len=hist_len.astype(np.int64), neg=neg,
                row_user=users.astype(np.int64), row_item=pos_items.astype(np.int64),
                V=V, n_items=n_items, n_users=n_users, T=T, S=S, n_cat=n_cat)


def build_reference_model(kind, world, D, embed, table_seed, dropout0=True, heads=8):
    """Mirror of Manager.__init__ order (loader/manager.py:139-153,294-326) on fake tables."""
    from loader.env import Env
    Env.device = torch.device("cpu")
    from loader.column_map import ColumnMap
    from loader.embedding_hub import EmbeddingHub
    from model.lego_config import LegoConfig
    from model.legommender import Legommender
    from model.operators.ada_operator import AdaOperator
    from model.operators.attention_operator import AttentionOperator
    from model.operators.cnn_operator import CNNOperator
    from model.predictors.dot_predictor import DotPredictor
    from loader.resampler import Resampler

    w = world
    glove_v = SizedVocab("glove", w["V"])
    cat_v = SizedVocab("category", w["n_cat"])
    item_v = SizedVocab("item_id", w["n_items"])
    user_v = SizedVocab("user_id", w["n_users"])
    item_rows = [{"item_id": i,
                  "title@glove": w["title_tok"][i, : w["title_len"][i]].tolist(),
                  "category": int(w["cat"][i])} for i in range(w["n_items"])]
    item_ut = FakeUT(item_rows, [Feat("item_id", item_v), Feat("title@glove", glove_v, w["T"]),
                                 Feat("category", cat_v)], "item_id")
    user_rows = [{"user_id": u, "history": list(w["hist"][u]), "neg": list(w["neg"][u])}
                 for u in range(w["n_users"])]
    user_ut = FakeUT(user_rows, [Feat("user_id", user_v), Feat("history", item_v, w["S"]),
                                 Feat("neg", item_v, 100)], "user_id")
    inter_rows = [{"index": r, "user_id": int(w["row_user"][r]), "item_id": int(w["row_item"][r]),
                   "click": 1, "history": list(w["hist"][w["row_user"][r]]),
                   "neg": list(w["neg"][w["row_user"][r]])} for r in range(len(w["row_user"]))]
    inter_ut = FakeUT(inter_rows, [Feat("index", SizedVocab("index", len(inter_rows))),
                                   Feat("user_id", user_v), Feat("item_id", item_v),
                                   Feat("click", SizedVocab("click", 2)),
                                   Feat("history", item_v, w["S"]), Feat("neg", item_v, 100)], "index")

    p = 0.0 if dropout0 else 0.1
    if kind == "naml":
        lc = LegoConfig(hidden_size=D, item_hidden_size=D, neg_count=4,
                        user_config={"inputer_config": {"use_cls_token": False, "use_sep_token": False}},
                        item_config={"dropout": p, "kernel_size": 3})
        lc.set_component_classes(CNNOperator, AdaOperator, DotPredictor)
    else:
        lc = LegoConfig(hidden_size=D, item_hidden_size=D, neg_count=4,
                        item_config={"num_attention_heads": heads, "attention_dropout": p,
                                     "inputer_config": {"use_cls_token": False, "use_sep_token": True}},
                        user_config={"num_attention_heads": heads, "attention_dropout": p,
                                     "inputer_config": {"use_cls_token": False, "use_sep_token": False}})
        lc.set_component_classes(AttentionOperator, AttentionOperator, DotPredictor)
    lc.set_item_ut(item_ut, ["title@glove", "category"])
    lc.set_user_ut(user_ut, ["history"])
    lc.set_column_map(ColumnMap(item_col="item_id", user_col="user_id", history_col="history",
                                neg_col="neg", label_col="click", group_col="user_id"))
    eh = EmbeddingHub(embedding_dim=D, transformation="auto", transformation_dropout=p)
    if embed == "glove":
        path = "/tmp/_golden_glove_%d_%d.npy" % (table_seed, w["V"])
        np.save(path, glove_table(table_seed, w["V"]))
        eh.load_pretrained_embedding(path, vocab_name="glove", frozen=True)
    eh.register_ut(item_ut, ["title@glove", "category"])
    lc.set_embedding_hub(eh)
    lc.build_components()
    lc.register_inputer_vocabs()
    model = Legommender(lc)
    resampler = Resampler(lc)
    return model, resampler, inter_ut, lc


def reference_batch(resampler, inter_ut, rows):
    """DataSet.__getitem__ + default_collate for the given interaction rows (train phase)."""
    from loader.data_set import DataSet
    from torch.utils.data import default_collate
    ds = DataSet(inter_ut, resampler)
    return default_collate([ds[r] for r in rows])


def flat_batch(kind, batch, world):
    """Recover the flat id view (candidate ids / history ids) of a nested reference batch.

    The reference stacks per-item content tensors; the item id of each stacked entry is
    recovered by matching its title/category tensors against the item table.
    """
    w = world
    key = {}
    for i in range(w["n_items"]):
        key.setdefault((tuple(w["title_tok"][i].tolist()), int(w["cat"][i])), i)

    def ids_of(nested):
        ii = nested["input_ids"]
        t = ii["title@glove"]
        c = ii["category"]
        B, C = t.shape[:2]
        out = np.zeros((B, C), dtype=np.int64)
        for b in range(B):
            for j in range(C):
                if kind == "naml":
                    tt = tuple(t[b, j].tolist())
                    cc = int(c[b, j, 0])
                else:  # concat layout: title prefix then -1; category sits after the first SEP
                    row = t[b, j].tolist()
                    L = sum(1 for x in row if x >= 0)
                    tt = tuple(row[:L] + [-1] * (w["T"] - L))
                    cc = int(c[b, j][c[b, j] >= 0][0])
                out[b, j] = key[(tt, cc)]
        return out
    cand = ids_of(batch["item_id"])
    hist = ids_of(batch["history"])
    mask = batch["__clicks_mask__"].numpy()
    hist_len = mask.sum(1)
    return cand, hist * mask, hist_len


def state_np(model):
    return {"param::" + k: v.detach().numpy().copy() for k, v in model.state_dict().items()}


def world_np(world):
    w = world
    S = w["S"]
    hist = np.zeros((w["n_users"], S), dtype=np.int64)
    for u, h in enumerate(w["hist"]):
        hist[u, : len(h)] = h
    return {"title_tok": w["title_tok"], "title_len": w["title_len"], "cat": w["cat"],
            "user_hist": hist, "user_hist_len": w["hist_len"]}


def model_fixture(name, kind, embed, D, V, n_items, n_users, B, seed, heads=8):
    from loader.env import Env
    torch.manual_seed(seed)
    random.seed(seed)
    np.random.seed(seed)
    world = make_world(seed, V, n_items, n_users, n_rows=4 * B)
    model, resampler, inter_ut, lc = build_reference_model(kind, world, D, embed, table_seed=seed + 1,
                                                           heads=heads)
    # make the (zero-init / tiny) biases non-trivial so bias handling is pinned too
    with torch.no_grad():
        for n, p_ in model.named_parameters():
            if p_.requires_grad and n.endswith("bias"):
                p_.add_(torch.randn_like(p_) * 0.05)
    Env.train()
    model.train()
    batch = reference_batch(resampler, inter_ut, list(range(B)))
    cand, hist, hist_len = flat_batch(kind, batch, world)
    assert (cand[:, 0] == world["row_item"][:B]).all()
    batch2 = clone_batch(batch)   # forward rewrites pad ids in place (`seq *= mask`): clone first

    # train-mode loss + grads (all dropouts are 0 in this model -> deterministic)
    model.zero_grad()
    loss = model(batch=batch)
    loss.backward()
    grads = {"grad::" + n: p_.grad.detach().numpy().copy()
             for n, p_ in model.named_parameters() if p_.requires_grad and p_.grad is not None}
    # eval-mode logits: batch built in train phase (negatives present), scored in test phase
    Env.test()
    model.eval()
    with torch.no_grad():
        logits = model(batch=batch2).numpy().copy()
    Env.train()
    out = {}
    out.update(state_np(model))
    out.update(grads)
    out.update(world_np(world))
    out.update({"cand": cand, "hist": hist, "hist_len": hist_len,
                "logits": logits, "loss": np.float32(loss.item())})
    meta = dict(kind=kind, embed=embed, D=D, V=V, n_items=n_items, n_users=n_users, B=B, seed=seed,
                heads=heads, table_seed=seed + 1, torch=torch.__version__,
                note="glove table = glove_table(table_seed, V) (regenerate, not stored)"
                if embed == "glove" else "trainable token table stored in params")
    if embed == "glove":   # do not store the big frozen table; it is regenerated from its seed
        out.pop("param::embedding_vocab_table.glove.embedding.weight")
    out["meta"] = np.frombuffer(json.dumps(meta).encode(), dtype=np.uint8)
    np.savez_compressed(os.path.join(OUT, name + ".npz"), **out)
    print(name, "loss", float(loss), "logits[0]", logits[0])


def clone_batch(batch):
    if isinstance(batch, dict):
        return {k: clone_batch(v) for k, v in batch.items()}
    return batch.clone()


# --------------------------------------------------------------------------- per-op fixtures
def op_fixtures(seed=7):
    from loader.env import Env
    Env.device = torch.device("cpu")
    from model.common.attention import AdditiveAttention
    torch.manual_seed(seed)
    out = {}
    # (1) additive attention: random 0/1 mask incl. an all-masked row and a full row
    n, L, D, A = 6, 31, 64, 48
    att = AdditiveAttention(embed_dim=D, hidden_size=A)
    with torch.no_grad():
        att.encoder[0].bias.add_(torch.randn(A) * 0.1)
    x = torch.randn(n, L, D, requires_grad=True)
    mask = (torch.rand(n, L) > 0.4).long()
    mask[0] = 0
    mask[1] = 1
    y = att(x, mask)
    gy = torch.randn_like(y)
    y.backward(gy)
    out.update({"add.x": x.detach().numpy(), "add.mask": mask.numpy(), "add.y": y.detach().numpy(),
                "add.gy": gy.numpy(), "add.gx": x.grad.numpy(),
                "add.W1": att.encoder[0].weight.detach().numpy(), "add.b1": att.encoder[0].bias.detach().numpy(),
                "add.w2": att.encoder[2].weight.detach().numpy(),
                "add.gW1": att.encoder[0].weight.grad.numpy(), "add.gb1": att.encoder[0].bias.grad.numpy(),
                "add.gw2": att.encoder[2].weight.grad.numpy()})
    # (2) nn.MultiheadAttention as AttentionOperator uses it (key_padding_mask, batch_first), train mode p=0
    n, L, D, H = 5, 33, 64, 8
    mha = torch.nn.MultiheadAttention(embed_dim=D, num_heads=H, dropout=0.0, batch_first=True)
    with torch.no_grad():
        mha.in_proj_bias.add_(torch.randn(3 * D) * 0.1)
        mha.out_proj.bias.add_(torch.randn(D) * 0.1)
    x = torch.randn(n, L, D, requires_grad=True)
    lens = torch.tensor([33, 1, 7, 20, 12])
    mask = (torch.arange(L)[None, :] < lens[:, None]).long()
    mha.train()
    y, _ = mha(query=x, key=x, value=x, key_padding_mask=(1 - mask).bool(), need_weights=False)
    gy = torch.randn_like(y) * mask[..., None]     # rows beyond the mask are never consumed downstream
    y.backward(gy)
    out.update({"mha.x": x.detach().numpy(), "mha.mask": mask.numpy(), "mha.y": y.detach().numpy(),
                "mha.gy": gy.numpy(), "mha.gx": x.grad.numpy(), "mha.heads": np.int64(H),
                "mha.in_w": mha.in_proj_weight.detach().numpy(), "mha.in_b": mha.in_proj_bias.detach().numpy(),
                "mha.out_w": mha.out_proj.weight.detach().numpy(), "mha.out_b": mha.out_proj.bias.detach().numpy(),
                "mha.gin_w": mha.in_proj_weight.grad.numpy(), "mha.gin_b": mha.in_proj_bias.grad.numpy(),
                "mha.gout_w": mha.out_proj.weight.grad.numpy(), "mha.gout_b": mha.out_proj.bias.grad.numpy()})
    # (3) dot predictor + CE with label 0 (legommender.py:254,263; dot_predictor.py:10)
    B, C, D = 7, 5, 64
    u = torch.randn(B, D, requires_grad=True)
    it = torch.randn(B, C, D, requires_grad=True)
    s = torch.sum(u.unsqueeze(1).repeat(1, C, 1).view(-1, D) * it.view(-1, D), dim=-1).view(B, C)
    loss = torch.nn.CrossEntropyLoss()(s, torch.zeros(B, dtype=torch.long))
    loss.backward()
    out.update({"dot.u": u.detach().numpy(), "dot.i": it.detach().numpy(), "dot.s": s.detach().numpy(),
                "dot.loss": np.float32(loss.item()), "dot.gu": u.grad.numpy(), "dot.gi": it.grad.numpy()})
    # (4) Adam + HF linear schedule with warm-up 0: 3-step trajectory (base_lego.py:201-223)
    from transformers import get_linear_schedule_with_warmup
    p_ = torch.nn.Parameter(torch.randn(37))
    opt = torch.optim.Adam([p_], lr=1e-3)
    sch = get_linear_schedule_with_warmup(opt, num_warmup_steps=0, num_training_steps=10)
    traj, gs = [p_.detach().numpy().copy()], []
    for step in range(3):
        g = torch.randn(37)
        gs.append(g.numpy().copy())
        p_.grad = g.clone()
        opt.step()
        sch.step()
        traj.append(p_.detach().numpy().copy())
    out.update({"adam.traj": np.stack(traj), "adam.g": np.stack(gs), "adam.lr": np.float32(1e-3),
                "adam.total": np.int64(10)})
    np.savez_compressed(os.path.join(OUT, "ops.npz"), **out)
    print("ops.npz written")


def metric_fixture(seed=11):
    from utils.metrics import MetricPool
    rs = np.random.RandomState(seed)
    n_groups, rows = 40, []
    for g in range(n_groups):
        k = rs.randint(2, 12)
        lab = np.zeros(k, dtype=np.int64)
        lab[rs.randint(0, k)] = 1
        if k > 3 and rs.rand() < 0.5:
            lab[rs.randint(0, k)] = 1
        if lab.sum() == k:
            lab[0] = 0
        sc = rs.randn(k).astype(np.float32)
        if g % 5 == 0:                      # ties
            sc[: k // 2] = sc[0]
        rows += [(g, int(l), float(s)) for l, s in zip(lab, sc)]
    groups = np.array([r[0] for r in rows])
    labels = np.array([r[1] for r in rows])
    scores = np.array([r[2] for r in rows], dtype=np.float32)
    names = ["GAUC", "MRR", "NDCG@1", "NDCG@5", "NDCG@10"]
    pool = MetricPool.parse(names)
    res = pool.calculate(scores.tolist(), labels.tolist(), groups.tolist())
    np.savez_compressed(os.path.join(OUT, "metrics.npz"), groups=groups, labels=labels, scores=scores,
                        names=np.array(names), values=np.array([float(res[n]) for n in names]))
    print("metrics", res)


FULL_METRICS = ["GAUC", "MRR", "MRR0", "NDCG@1", "NDCG@5", "NDCG@10", "HitRatio@1", "HitRatio@5", "HitRatio@10",
                "Recall@1", "Recall@5", "Recall@10", "LRAP", "AUC", "LogLoss", "F1"]


def metric_fixture_full(seed=12):
    """every metric of MetricPool.metric_list on probabilities in (0, 1): groups of 2..150 rows, 1..4 positives, ties
    (including ties between positives and negatives and an all-equal group) -> metrics_full.npz"""
    from utils.metrics import MetricPool
    rs = np.random.RandomState(seed)
    groups, labels, scores = [], [], []
    for g in range(60):
        k = int(rs.randint(2, 12)) if g % 6 else int(rs.randint(40, 151))
        lab = np.zeros(k, dtype=np.int64)
        lab[rs.choice(k, size=min(int(rs.randint(1, 5)), k - 1), replace=False)] = 1
        sc = (1.0 / (1.0 + np.exp(-rs.randn(k)))).astype(np.float32)
        if g % 5 == 0:
            sc[: k // 2] = sc[0]
        if g % 7 == 3:
            sc = np.round(sc, 1)             # many small tie groups
        if g == 11:
            sc[:] = 0.5
        groups += [1000 - g] * k             # group ids are not sorted and not dense
        labels += lab.tolist()
        scores += sc.tolist()
    perm = rs.permutation(len(groups))       # rows of one group are not contiguous
    groups, labels = np.array(groups)[perm], np.array(labels)[perm]
    scores = np.array(scores, dtype=np.float32)[perm]
    pool = MetricPool.parse(FULL_METRICS)
    res = pool.calculate(scores.tolist(), labels.tolist(), groups.tolist())
    names = [str(m) for m in pool.metrics]
    np.savez_compressed(os.path.join(OUT, "metrics_full.npz"), groups=groups, labels=labels, scores=scores,
                        names=np.array(names), values=np.array([float(res[n]) for n in names]))
    print("metrics_full", res)


def main():
    assert os.path.isdir(REF), "the reference is only present in the build container"
    install_stubs()
    torch.set_num_threads(8)
    if sys.argv[1:] == ["metrics_full"]:
        return metric_fixture_full()
    op_fixtures()
    metric_fixture()
    metric_fixture_full()
    # small full-model pins (tiny shapes)
    model_fixture("naml_glove_d64", "naml", "glove", D=64, V=500, n_items=120, n_users=40, B=8, seed=2023)
    model_fixture("nrms_null_d64", "nrms", "null", D=64, V=500, n_items=120, n_users=40, B=8, seed=2024)
    model_fixture("nrms_glove_d64", "nrms", "glove", D=64, V=500, n_items=120, n_users=40, B=6, seed=2025)
    # BASELINE config 1 shape: NAML hidden=64 bs=32 GloVe (CPU-runnable case)
    model_fixture("naml_glove_cfg1", "naml", "glove", D=64, V=2000, n_items=400, n_users=120, B=32, seed=2026)
    # BASELINE config 2 hidden size at a reduced batch (hidden=256)
    model_fixture("naml_glove_d256", "naml", "glove", D=256, V=800, n_items=150, n_users=40, B=4, seed=2027)


if __name__ == "__main__":
    main()
