"""MIND-small-shaped synthetic world: there is no MIND / GloVe on disk, so the
bench and the smoke test use seeded tables of the same shapes and raggedness.

    items   n_items = 65 238, title length ~ U[5,30] tokens (Zipf over V = 400 000), 18 categories
    users   n_users = 91 935, history length ~ clipped geometric (mean ~ 20) in [1,50],
            true-negative list length ~ U[0,100]
    train   208 238 positive rows (user, clicked item)
    GloVe   N(0, 0.4^2) [V, 300] fp32 (frozen)

numpy (seeded) builds the integer tables on the host once; they are then resident in HBM.
"""
from __future__ import annotations

import numpy as np
import torch

MIND_SMALL = dict(n_items=65238, n_users=91935, n_rows=208238, V=400000, T=30, S=50, n_cat=18, neg_cap=100)


def make_world(seed=2023, n_items=65238, n_users=91935, n_rows=208238, V=400000, T=30, S=50, n_cat=18,
               neg_cap=100, hist_mean=20.0):
    rs = np.random.RandomState(seed)
    title_len = rs.randint(5, T + 1, size=n_items).astype(np.int32)
    z = np.minimum(rs.zipf(1.2, size=(n_items, T)) - 1, V - 1).astype(np.int32)
    title_tok = np.where(np.arange(T)[None, :] < title_len[:, None], z, -1).astype(np.int32)
    cat = rs.randint(0, n_cat, size=n_items).astype(np.int32)
    hist_len = np.clip(rs.geometric(1.0 / hist_mean, size=n_users), 1, S).astype(np.int32)
    user_hist = (rs.randint(0, n_items, size=(n_users, S)) * (np.arange(S)[None, :] < hist_len[:, None])).astype(np.int32)
    neg_len = rs.randint(0, neg_cap + 1, size=n_users).astype(np.int32)
    neg_list = rs.randint(0, n_items, size=(n_users, neg_cap)).astype(np.int32)
    row_user = rs.randint(0, n_users, size=n_rows).astype(np.int32)
    row_item = rs.randint(0, n_items, size=n_rows).astype(np.int32)
    return dict(title_tok=title_tok, title_len=title_len, cat=cat, user_hist=user_hist, user_hist_len=hist_len,
                neg_list=neg_list, neg_len=neg_len, row_user=row_user, row_item=row_item,
                n_items=n_items, n_users=n_users, n_rows=n_rows, V=V, T=T, S=S, n_cat=n_cat, neg_cap=neg_cap)


def glove_like(V, E0=300, seed=2024, device="cpu"):
    g = torch.Generator(device="cpu").manual_seed(seed)
    if str(device) == "cpu":
        return torch.randn(V, E0, generator=g) * 0.4
    # generate on the device in chunks (480 MB at V = 400 000): data generation, not the data path
    gd = torch.Generator(device=device).manual_seed(seed)
    return torch.randn(V, E0, generator=gd, device=device) * 0.4


def init_naml_params(D=256, A=256, E0=300, V=400000, n_cat=18, seed=2023, glove=None):
    """PyTorch-default initialisation of exactly the reference's NAML modules (state_dict key names of
    SURVEY.md section 8b): nn.Linear / nn.Conv1d / nn.Embedding defaults."""
    torch.manual_seed(seed)
    nn = torch.nn
    lin_p, cat_e = nn.Linear(E0, D), nn.Embedding(n_cat, D)
    cnn, lin_i = nn.Conv1d(D, D, 3, padding="same"), nn.Linear(D, D)
    a0, a2 = nn.Linear(D, A), nn.Linear(A, 1, bias=False)
    u0, u2 = nn.Linear(D, A), nn.Linear(A, 1, bias=False)
    P = {
        "embedding_vocab_table.glove.embedding.weight": glove if glove is not None else glove_like(V, E0),
        "embedding_vocab_table.glove.linear.weight": lin_p.weight, "embedding_vocab_table.glove.linear.bias": lin_p.bias,
        "embedding_vocab_table.category.weight": cat_e.weight,
        "item_op.cnn.weight": cnn.weight, "item_op.cnn.bias": cnn.bias,
        "item_op.linear.weight": lin_i.weight, "item_op.linear.bias": lin_i.bias,
        "item_op.additive_attention.encoder.0.weight": a0.weight, "item_op.additive_attention.encoder.0.bias": a0.bias,
        "item_op.additive_attention.encoder.2.weight": a2.weight,
        "user_op.additive_attention.encoder.0.weight": u0.weight, "user_op.additive_attention.encoder.0.bias": u0.bias,
        "user_op.additive_attention.encoder.2.weight": u2.weight,
    }
    return {k: v.detach().clone().float().contiguous() for k, v in P.items()}


def init_nrms_params(D=256, A=256, V=400000, n_cat=18, heads=8, seed=2023, glove=None, E0=300):
    """Reference NRMS modules (AttentionOperator x2).  glove=None -> trainable [V,D] token table (embed/null)."""
    torch.manual_seed(seed)
    nn = torch.nn
    P = {}
    if glove is not None:
        lin_p = nn.Linear(E0, D)
        P["embedding_vocab_table.glove.embedding.weight"] = glove
        P["embedding_vocab_table.glove.linear.weight"] = lin_p.weight
        P["embedding_vocab_table.glove.linear.bias"] = lin_p.bias
    else:
        P["embedding_vocab_table.glove.weight"] = nn.Embedding(V, D).weight
    P["embedding_vocab_table.category.weight"] = nn.Embedding(n_cat, D).weight
    P["embedding_vocab_table.__cat_inputer_special_ids.weight"] = nn.Embedding(3, D).weight
    for pre in ("item_op.", "user_op."):
        mha = nn.MultiheadAttention(D, heads, dropout=0.1, batch_first=True)
        lin, a0, a2 = nn.Linear(D, D), nn.Linear(D, A), nn.Linear(A, 1, bias=False)
        P[pre + "multi_head_attention.in_proj_weight"] = mha.in_proj_weight
        P[pre + "multi_head_attention.in_proj_bias"] = mha.in_proj_bias
        P[pre + "multi_head_attention.out_proj.weight"] = mha.out_proj.weight
        P[pre + "multi_head_attention.out_proj.bias"] = mha.out_proj.bias
        P[pre + "linear.weight"] = lin.weight
        P[pre + "linear.bias"] = lin.bias
        P[pre + "additive_attention.encoder.0.weight"] = a0.weight
        P[pre + "additive_attention.encoder.0.bias"] = a0.bias
        P[pre + "additive_attention.encoder.2.weight"] = a2.weight
    return {k: v.detach().clone().float().contiguous() for k, v in P.items()}


def make_learnable_world(seed=0, n_items=1200, n_users=900, n_rows=9600, V=3000, T=16, S=20, n_cat=18, neg_cap=20,
                         p_pref=0.8, p_topic=0.5, pool=30, n_dev_users=400, dev_neg=8):
    """A small world with PLANTED signal (the MIND-shaped `make_world` is label-free: its AUC is 0.5 whatever is trained).

    Every user prefers two categories.  Clicked items (history, train positives, dev positives) come from the preferred
    categories with probability `p_pref`, else from anywhere; impressed-but-not-clicked items (the true-negative lists, the
    dev negatives) come from the OTHER categories with the same probability.  An item shows its category twice: the
    category column, and title tokens drawn with probability `p_topic` from a pool of `pool` token ids owned by the category
    (ids [100 + c * pool, 100 + (c + 1) * pool)), the rest Zipf over the vocabulary.  So a model that learns the match
    between a history's categories / topic tokens and a candidate's separates clicks from non-clicks; the ceiling is well
    below 1 (a fifth of each side is noise).  `valid` = one group per dev user: 2 clicked + `dev_neg` non-clicked rows
    (both classes in every group).  Legacy numpy RandomState: identical on every machine."""
    rs = np.random.RandomState(seed)
    cat = rs.randint(0, n_cat, size=n_items).astype(np.int32)
    cat[:n_cat] = np.arange(n_cat)                                   # every category has an item
    title_len = rs.randint(4, T + 1, size=n_items).astype(np.int32)
    generic = np.minimum(rs.zipf(1.2, size=(n_items, T)) - 1, V - 1)
    topic = 100 + cat[:, None] * pool + rs.randint(0, pool, size=(n_items, T))
    tok = np.where(rs.rand(n_items, T) < p_topic, topic, generic)
    title_tok = np.where(np.arange(T)[None, :] < title_len[:, None], tok, -1).astype(np.int32)
    by_cat = [np.flatnonzero(cat == c) for c in range(n_cat)]
    pref = np.stack([rs.permutation(n_cat)[:2] for _ in range(n_users)])              # [n_users, 2]
    liked = np.zeros((n_users, n_cat), dtype=bool)
    liked[np.arange(n_users)[:, None], pref] = True

    def draw(users, want_liked):
        """one item per entry of `users`: from the (dis)liked categories w.p. p_pref, else uniform over all items"""
        out = rs.randint(0, n_items, size=len(users))
        follow = rs.rand(len(users)) < p_pref
        for i in np.flatnonzero(follow):
            u = users[i]
            cs = np.flatnonzero(liked[u] == want_liked)
            pool_items = by_cat[cs[rs.randint(len(cs))]]
            out[i] = pool_items[rs.randint(len(pool_items))]
        return out.astype(np.int32)

    hist_len = np.clip(rs.geometric(1.0 / 8.0, size=n_users), 1, S).astype(np.int32)
    flat_u = np.repeat(np.arange(n_users), S)
    user_hist = (draw(flat_u, True).reshape(n_users, S) * (np.arange(S)[None, :] < hist_len[:, None])).astype(np.int32)
    neg_len = rs.randint(0, neg_cap + 1, size=n_users).astype(np.int32)
    neg_list = draw(np.repeat(np.arange(n_users), neg_cap), False).reshape(n_users, neg_cap)
    row_user = rs.randint(0, n_users, size=n_rows).astype(np.int32)
    row_item = draw(row_user, True)
    dev_users = rs.permutation(n_users)[:n_dev_users]
    per = 2 + dev_neg
    vu = np.repeat(dev_users, per).astype(np.int32)
    vl = np.tile(np.array([1, 1] + [0] * dev_neg, dtype=np.int64), n_dev_users)
    vi = np.where(vl == 1, draw(vu, True), draw(vu, False)).astype(np.int32)
    return dict(title_tok=title_tok, title_len=title_len, cat=cat, user_hist=user_hist, user_hist_len=hist_len,
                neg_list=neg_list, neg_len=neg_len, row_user=row_user, row_item=row_item,
                n_items=n_items, n_users=n_users, n_rows=n_rows, V=V, T=T, S=S, n_cat=n_cat, neg_cap=neg_cap,
                valid=dict(user=vu, item=vi, label=vl))


def glove_table_np(seed, V, E0=300):
    """the frozen [V, 300] table of the small fixtures, regenerated from its seed (legacy RandomState: machine-independent)"""
    return (np.random.RandomState(seed).standard_normal((V, E0)) * 0.4).astype(np.float32)
