"""Raw MIND -> this package's npz tables (reference: processor/mind_processor.py:30-227,
processor/base_processor.py:265-373, embedder/glove_embedder.py:75-125).

    python -m legommenders_amd.process_mind --mind_dir <dir with train/ dev/> --glove <glove.6B.300d.txt> --out data/mind

Semantics kept from the reference: items = train+dev news.tsv de-duplicated by nid; users = first occurrence per uid of
train+dev behaviors with the history filtered to known news, users with an empty history dropped; interactions =
exploded "nid-label" tokens of users that survived; 10 % of the TRAIN users (seeded shuffle) become the validation split,
dev/behaviors.tsv is the test split; per-user true-negative lists from train+valid label-0 rows (first NEG_TRUNCATE=100);
positive train rows only are training samples (loader/manager.py:331-347).  The reference stores UniTok directories (a
third-party format that is not vendored); here everything lands in five .npz files + embeddings/glove.npy.
Titles are lower-cased and split with a simple regex word/punctuation tokenizer; words outside the GloVe vocabulary are
dropped (the reference's GloVeTokenizer relies on nltk, which is not available here: tokenisation parity is unpinned).
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


def build(mind_dir: str, glove_path: str, out_dir: str, seed: int = 2023) -> Dict[str, int]:
    vocab, vectors = load_glove(glove_path)
    news = read_news(os.path.join(mind_dir, "train", "news.tsv"))
    for k, v in read_news(os.path.join(mind_dir, "dev", "news.tsv")).items():
        news.setdefault(k, v)
    nid = {k: i for i, k in enumerate(news)}
    cats: Dict[str, int] = {}
    n_items = len(news)
    title_tok = -np.ones((n_items, TITLE_LEN), dtype=np.int32)
    title_len = np.zeros(n_items, dtype=np.int32)
    cat = np.zeros(n_items, dtype=np.int32)
    for k, (c, title) in news.items():
        i = nid[k]
        cat[i] = cats.setdefault(c, len(cats))
        toks = [vocab[t] for t in _TOK.findall(title.lower()) if t in vocab][:TITLE_LEN]
        title_tok[i, : len(toks)] = toks
        title_len[i] = len(toks)
    train_b = read_behaviors(os.path.join(mind_dir, "train", "behaviors.tsv"))
    dev_b = read_behaviors(os.path.join(mind_dir, "dev", "behaviors.tsv"))
    users: Dict[str, List[int]] = {}
    for uid, hist, _ in train_b + dev_b:
        if uid not in users:
            h = [nid[n] for n in hist if n in nid]
            if h:
                users[uid] = h
    uidx = {u: i for i, u in enumerate(users)}

    def explode(beh):
        return [(uidx[u], nid[n], int(l)) for u, _, preds in beh if u in uidx for n, l in preds if n in nid]
    train, test = explode(train_b), explode(dev_b)
    tr_users = list(dict.fromkeys(u for u, _, _ in train))
    random.Random(seed).shuffle(tr_users)
    valid_users = set(tr_users[: int(len(tr_users) * 0.1)])
    valid = [r for r in train if r[0] in valid_users]
    train = [r for r in train if r[0] not in valid_users]
    n_users = len(users)
    user_hist = np.zeros((n_users, HIST_LEN), dtype=np.int32)
    user_hist_len = np.zeros(n_users, dtype=np.int32)
    for u, h in users.items():
        h = h[:HIST_LEN]
        user_hist[uidx[u], : len(h)] = h
        user_hist_len[uidx[u]] = len(h)
    neg_list = np.zeros((n_users, NEG_TRUNCATE), dtype=np.int32)
    neg_len = np.zeros(n_users, dtype=np.int32)
    for u, n, l in train + valid:
        if l == 0 and neg_len[u] < NEG_TRUNCATE:
            neg_list[u, neg_len[u]] = n
            neg_len[u] += 1
    os.makedirs(out_dir, exist_ok=True)
    np.savez_compressed(os.path.join(out_dir, "items.npz"), title_tok=title_tok, title_len=title_len, cat=cat,
                        vocab_size=np.int64(len(vocab)))
    np.savez_compressed(os.path.join(out_dir, "users.npz"), user_hist=user_hist, user_hist_len=user_hist_len,
                        neg_list=neg_list, neg_len=neg_len)
    pos = [(u, n) for u, n, l in train if l == 1]
    np.savez_compressed(os.path.join(out_dir, "train.npz"), row_user=np.array([p[0] for p in pos], dtype=np.int32),
                        row_item=np.array([p[1] for p in pos], dtype=np.int32))
    for name, rows in (("valid", valid), ("test", test)):
        a = np.array(rows, dtype=np.int64).reshape(-1, 3)
        np.savez_compressed(os.path.join(out_dir, name + ".npz"), user=a[:, 0], item=a[:, 1], label=a[:, 2])
    emb_dir = os.path.join(os.path.dirname(out_dir.rstrip("/")) or ".", "embeddings")
    os.makedirs(emb_dir, exist_ok=True)
    np.save(os.path.join(emb_dir, "glove.npy"), vectors)
    return dict(items=n_items, users=n_users, train=len(pos), valid=len(valid), test=len(test), categories=len(cats))


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--mind_dir", required=True)
    ap.add_argument("--glove", required=True)
    ap.add_argument("--out", default="data/mind")
    ap.add_argument("--seed", type=int, default=2023)
    a = ap.parse_args()
    print(build(a.mind_dir, a.glove, a.out, a.seed))
