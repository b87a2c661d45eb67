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
