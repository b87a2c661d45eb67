"""Raw MIND -> this package's npz tables (reference: processor/mind_processor.py:30-227,
processor/base_processor.py:265-373, embedder/glove_embedder.py:75-125).

    python -m legommenders_amd.process_mind --mind_dir <dir with train/ dev/> --glove <glove.6B.300d.txt> --out data/mind

Semantics of the reference, pinned to a run of its own `MINDProcessor.load()` on tests/golden/mind_raw (fixture
tests/golden/mind_tables.json, generator tests/golden/make_golden_mind.py, test tests/test_process_mind.py): items =
train+dev news.tsv de-duplicated by nid; users = first occurrence per uid of train+dev behaviors with the history
filtered to known news, users with an empty history dropped; interactions = exploded "nid-label" tokens of users that
survived; 10 % of the TRAIN users (seeded shuffle) become the validation split, dev/behaviors.tsv is the test split;
per-user true-negative lists from train+valid label-0 rows (first NEG_TRUNCATE=100); users / items that nothing refers to
are removed before ids are assigned; positive train rows only are training samples (loader/manager.py:331-347).  The
reference stores UniTok directories (a third-party format that is not vendored); here everything lands in five .npz
files + embeddings/glove.npy.  Titles are lower-cased and split with a simple regex word/punctuation tokenizer; words
outside the GloVe vocabulary are dropped (the reference's GloVeTokenizer relies on nltk, which is not available here:
the TOKEN IDS of titles are the one part of these tables whose parity stays unpinned).
"""
from __future__ import annotations

import argparse
import os
import random
import re
from typing import Dict, List

import numpy as np

NEG_TRUNCATE = 100
TITLE_LEN, HIST_LEN = 30, 50          # config/data/mind.yaml:5-10 (truncate to the FIRST n entries)
_TOK = re.compile(r"[a-z0-9]+|[^\sa-z0-9]")


def load_glove(path: str):
    words, vecs = [], []
    with open(path, encoding="utf-8") as f:
        for line in f:
            parts = line.rstrip().split(" ")
            words.append(parts[0])
            vecs.append(np.asarray(parts[1:], dtype=np.float32))
    return {w: i for i, w in enumerate(words)}, np.stack(vecs)


def read_news(path):
    rows = {}
    with open(path, encoding="utf-8") as f:
        for line in f:
            p = line.rstrip("\n").split("\t")
            if p[0] not in rows:
                rows[p[0]] = (p[1], p[3])          # category, title
    return rows


def read_behaviors(path):
    out = []
    with open(path, encoding="utf-8") as f:
        for line in f:
            p = line.rstrip("\n").split("\t")
            out.append((p[1], p[3].split(), [t.split("-") for t in p[4].split()]))
    return out


def build_tables(mind_dir: str, seed: int = 2023):
    """The reference's raw -> table logic up to (not including) tokenisation, on plain python containers:
    `MINDProcessor.load_items / load_users / load_interactions` (processor/mind_processor.py:98-205), the negative-list
    merge and the unused-user / unused-item compression of `BaseProcessor.load / generate` (processor/base_processor.py:
    238-280).  Returns ids in the reference's row order: an item's / user's index is its row in the compressed table
    (that is the index unitok's EntityTokenizer assigns when the table is tokenised).  Pinned to a run of the reference
    itself: tests/golden/mind_tables.json."""
    news = read_news(os.path.join(mind_dir, "train", "news.tsv"))
    for k, v in read_news(os.path.join(mind_dir, "dev", "news.tsv")).items():
        news.setdefault(k, v)                                    # concat + drop_duplicates(nid): first occurrence wins
    train_b = read_behaviors(os.path.join(mind_dir, "train", "behaviors.tsv"))
    dev_b = read_behaviors(os.path.join(mind_dir, "dev", "behaviors.tsv"))
    users: Dict[str, List[str]] = {}
    seen = set()
    for uid, hist, _ in train_b + dev_b:                         # first occurrence per uid; history filtered to known news;
        if uid in seen:                                          # users whose FIRST history is empty after filtering are dropped
            continue
        seen.add(uid)
        h = [n for n in hist if n in news]
        if h:
            users[uid] = h

    def explode(beh):
        return [(u, n, int(l)) for u, _, preds in beh if u in users for n, l in preds]
    train, test = explode(train_b), explode(dev_b)
    tr_users = list(dict.fromkeys(u for u, _, _ in train))       # train_df[uid].unique(): order of first appearance
    random.Random(seed).shuffle(tr_users)                        # == random.seed(seed); random.shuffle(users)
    valid_users = set(tr_users[: int(len(tr_users) * 0.1)])
    valid = [r for r in train if r[0] in valid_users]
    train = [r for r in train if r[0] not in valid_users]
    neg: Dict[str, List[str]] = {u: [] for u in users}
    for u, n, l in train + valid:                                # groupby(uid)[nid].apply(list) over concat(train, valid)
        if l == 0:
            neg[u].append(n)
    # compression (base_processor.py:253-280): users that occur in some interaction; items that occur in an interaction, a
    # history or the first NEG_TRUNCATE negatives of a kept user
    used_users = {u for u, _, _ in train + valid + test}
    users = {u: h for u, h in users.items() if u in used_users}
    used_items = {n for _, n, _ in train + valid + test}
    for u, h in users.items():
        used_items.update(h)
        used_items.update(neg[u][:NEG_TRUNCATE])
    news = {k: v for k, v in news.items() if k in used_items}
    return dict(news=news, users=users, neg={u: neg[u][:NEG_TRUNCATE] for u in users}, train=train, valid=valid, test=test)


def build(mind_dir: str, glove_path: str, out_dir: str, seed: int = 2023) -> Dict[str, int]:
    vocab, vectors = load_glove(glove_path)
    t = build_tables(mind_dir, seed)
    news, users = t["news"], t["users"]
    nid = {k: i for i, k in enumerate(news)}
    cats: Dict[str, int] = {}
    n_items = len(news)
    title_tok = -np.ones((n_items, TITLE_LEN), dtype=np.int32)
    title_len = np.zeros(n_items, dtype=np.int32)
    cat = np.zeros(n_items, dtype=np.int32)
    for k, (c, title) in news.items():
        i = nid[k]
        cat[i] = cats.setdefault(c, len(cats))                   # EntityTokenizer(vocab="category"): index by first appearance
        toks = [vocab[w] for w in _TOK.findall(title.lower()) if w in vocab][:TITLE_LEN]
        title_tok[i, : len(toks)] = toks
        title_len[i] = len(toks)
    uidx = {u: i for i, u in enumerate(users)}
    n_users = len(users)
    user_hist = np.zeros((n_users, HIST_LEN), dtype=np.int32)
    user_hist_len = np.zeros(n_users, dtype=np.int32)
    neg_list = np.zeros((n_users, NEG_TRUNCATE), dtype=np.int32)
    neg_len = np.zeros(n_users, dtype=np.int32)
    for u, h in users.items():
        h = [nid[n] for n in h][:HIST_LEN]
        user_hist[uidx[u], : len(h)] = h
        user_hist_len[uidx[u]] = len(h)
        ng = [nid[n] for n in t["neg"][u]]
        neg_list[uidx[u], : len(ng)] = ng
        neg_len[uidx[u]] = len(ng)
    rows = {name: [(uidx[u], nid[n], l) for u, n, l in t[name] if n in nid] for name in ("train", "valid", "test")}
    os.makedirs(out_dir, exist_ok=True)
    np.savez_compressed(os.path.join(out_dir, "items.npz"), title_tok=title_tok, title_len=title_len, cat=cat,
                        vocab_size=np.int64(len(vocab)), nid=np.array(list(news)), category=np.array(list(cats)))
    np.savez_compressed(os.path.join(out_dir, "users.npz"), user_hist=user_hist, user_hist_len=user_hist_len,
                        neg_list=neg_list, neg_len=neg_len, uid=np.array(list(users)))
    pos = [(u, n) for u, n, l in rows["train"] if l == 1]        # positive rows are the training samples (loader/manager.py:331-347)
    np.savez_compressed(os.path.join(out_dir, "train.npz"), row_user=np.array([p[0] for p in pos], dtype=np.int32),
                        row_item=np.array([p[1] for p in pos], dtype=np.int32))
    for name in ("valid", "test"):
        a = np.array(rows[name], dtype=np.int64).reshape(-1, 3)
        np.savez_compressed(os.path.join(out_dir, name + ".npz"), user=a[:, 0], item=a[:, 1], label=a[:, 2])
    emb_dir = os.path.join(os.path.dirname(out_dir.rstrip("/")) or ".", "embeddings")
    os.makedirs(emb_dir, exist_ok=True)
    np.save(os.path.join(emb_dir, "glove.npy"), vectors)
    return dict(items=n_items, users=n_users, train=len(pos), valid=len(rows["valid"]), test=len(rows["test"]), categories=len(cats))


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--mind_dir", required=True)
    ap.add_argument("--glove", required=True)
    ap.add_argument("--out", default="data/mind")
    ap.add_argument("--seed", type=int, default=2023)
    a = ap.parse_args()
    print(build(a.mind_dir, a.glove, a.out, a.seed))
