"""CPU oracle for the Legommenders two-tower hot path  --  TEST INFRASTRUCTURE ONLY.

This file is a from-scratch *functional restatement* (flat id tensors, fp32 torch-CPU
arithmetic) of the reference's `Legommender.forward` path for NAML and NRMS.  It is the
checker the HIP kernels are compared against and the `cpu_baseline` ("port") leg of
bench.py.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline may import it;
the product package `legommenders_amd` never does (tests/test_product_isolation.py).

Pinning: every function below is checked against golden vectors produced by importing the
REAL reference in the build container (tests/golden/make_golden.py -> tests/golden/*.npz;
tests/test_oracle_golden.py).  The reference has no tests of its own,
so those generated vectors are the pins.

Reference citations are relative to /root/reference.
"""
from __future__ import annotations

import math
from typing import Dict, Tuple

import numpy as np
import torch
import torch.nn.functional as F

EPS = float(torch.finfo(torch.float32).eps)   # model/common/attention.py:36
UNSET = -1                                      # loader/env.py:10
PAD, CLS, SEP = 0, 1, 2                         # model/inputer/concat_inputer.py:27-30


def _t(x, dtype=None):
    if isinstance(x, torch.Tensor):
        return x if dtype is None else x.to(dtype)
    t = torch.from_numpy(np.ascontiguousarray(x))
    return t if dtype is None else t.to(dtype)


# --------------------------------------------------------------------------- a6
def additive_attention(x, mask, W1, b1, w2):
    """AdditiveAttention.forward  (model/common/attention.py:23-38).

    a = w2 . tanh(W1 x + b1);  e = exp(a) * mask  (no max-subtraction);
    w = e / (sum(e) + eps);    out = sum_l w_l x_l.     x:[n,L,D] mask:[n,L] -> [n,D]
    """
    a = torch.tanh(F.linear(x, W1, b1))
    a = F.linear(a, w2).squeeze(-1)
    e = torch.exp(a) * mask.to(x.dtype)
    w = e / (e.sum(-1, keepdim=True) + EPS)
    return (x * w.unsqueeze(-1)).sum(1)


# --------------------------------------------------------------------------- a3 / a4
def glove_project(tok, table, lin_w, lin_b):
    """SimpleInputer.get_embeddings for a pre-trained column wrapped in `Transformation`
    (model/inputer/simple_inputer.py:55-63; loader/embedding_hub.py:95-96, eval / p=0).

    pad id -1 is rewritten to 0, row 0 is looked up and projected (+bias), and the result is
    zeroed at masked positions.  tok:[n,T] int64 with -1 pads -> ([n,T,D], mask[n,T]).
    """
    mask = (tok > UNSET).long()
    seq = tok * mask
    emb = F.linear(F.embedding(seq, table), lin_w, lin_b)
    return emb * mask.unsqueeze(-1).to(emb.dtype), mask


def table_lookup(tok, table):
    """Same inputer path with a plain trainable nn.Embedding (loader/embedding_hub.py:325-335)."""
    mask = (tok > UNSET).long()
    emb = F.embedding(tok * mask, table)
    return emb * mask.unsqueeze(-1).to(emb.dtype), mask


# --------------------------------------------------------------------------- a5
def cnn_operator(title_emb, title_mask, cat_emb, P, prefix="item_op."):
    """CNNOperator.forward (model/operators/cnn_operator.py:48-67), eval / dropout 0.

    title (len>1 column): Conv1d(k=3,'same') -> ReLU -> *mask;  category (len-1 column):
    Linear;  concat along L in `item.inputs` order (title, category);  additive pool.
    """
    y = F.conv1d(title_emb.permute(0, 2, 1), P[prefix + "cnn.weight"], P[prefix + "cnn.bias"], padding="same")
    y = torch.relu(y.permute(0, 2, 1)) * title_mask.unsqueeze(-1).to(title_emb.dtype)
    c = F.linear(cat_emb, P[prefix + "linear.weight"], P[prefix + "linear.bias"])
    out = torch.cat([y, c], dim=1)
    mask = torch.cat([title_mask, torch.ones_like(title_mask[:, :1])], dim=1)
    return additive_attention(out, mask,
                              P[prefix + "additive_attention.encoder.0.weight"],
                              P[prefix + "additive_attention.encoder.0.bias"],
                              P[prefix + "additive_attention.encoder.2.weight"])


# --------------------------------------------------------------------------- a8
def mhsa(x, mask, in_w, in_b, out_w, out_b, heads):
    """nn.MultiheadAttention(batch_first, key_padding_mask=(1-mask).bool(), need_weights=False)
    as called by AttentionOperator.forward (model/operators/attention_operator.py:49-55),
    restated explicitly: packed in-proj, per-head softmax(QK^T/sqrt(hd)) with masked keys = -inf,
    .V, out-proj.  attention-dropout = 0 (eval).  x:[n,L,D] mask:[n,L] -> [n,L,D]
    """
    n, L, D = x.shape
    hd = D // heads
    qkv = F.linear(x, in_w, in_b)
    q, k, v = qkv.split(D, dim=-1)
    q = q.view(n, L, heads, hd).transpose(1, 2)
    k = k.view(n, L, heads, hd).transpose(1, 2)
    v = v.view(n, L, heads, hd).transpose(1, 2)
    s = (q * (1.0 / math.sqrt(hd))) @ k.transpose(-1, -2)
    s = s.masked_fill((mask == 0)[:, None, None, :], float("-inf"))
    p = torch.softmax(s, dim=-1)
    o = (p @ v).transpose(1, 2).reshape(n, L, D)
    return F.linear(o, out_w, out_b)


def attention_operator(x, mask, P, prefix, heads):
    """AttentionOperator.forward (model/operators/attention_operator.py:46-59): MHSA -> Linear -> additive pool."""
    o = mhsa(x, mask, P[prefix + "multi_head_attention.in_proj_weight"],
             P[prefix + "multi_head_attention.in_proj_bias"],
             P[prefix + "multi_head_attention.out_proj.weight"],
             P[prefix + "multi_head_attention.out_proj.bias"], heads)
    o = F.linear(o, P[prefix + "linear.weight"], P[prefix + "linear.bias"])
    return additive_attention(o, mask,
                              P[prefix + "additive_attention.encoder.0.weight"],
                              P[prefix + "additive_attention.encoder.0.bias"],
                              P[prefix + "additive_attention.encoder.2.weight"])


# --------------------------------------------------------------------------- a3'
def concat_layout(title_tok, title_len, cat, use_sep=True):
    """ConcatInputer.sample_rebuilder (model/inputer/concat_inputer.py:58-87), vectorised over items.

    Compact `[title..., SEP, category, SEP, PAD...]`; three full-length id rows (title column,
    category column, special-token column) filled with -1 where the column is absent; the special
    column holds PAD(0) after the live prefix; mask = 1 on the live prefix.
    """
    n, T = title_tok.shape
    L = T + 1 + (2 if use_sep else 0)
    t_ids = torch.full((n, L), UNSET, dtype=torch.long)
    c_ids = torch.full((n, L), UNSET, dtype=torch.long)
    s_ids = torch.full((n, L), UNSET, dtype=torch.long)
    ar = torch.arange(L)[None, :]
    tl = title_len[:, None]
    t_ids[:, :T] = title_tok
    t_ids = torch.where(ar < tl, t_ids, torch.full_like(t_ids, UNSET))
    rows = torch.arange(n)
    if use_sep:
        s_ids[rows, title_len] = SEP
        c_ids[rows, title_len + 1] = cat
        s_ids[rows, title_len + 2] = SEP
        live = title_len + 3
    else:
        c_ids[rows, title_len] = cat
        live = title_len + 1
    s_ids = torch.where(ar >= live[:, None], torch.full_like(s_ids, PAD), s_ids)
    mask = (ar < live[:, None]).long()
    return t_ids, c_ids, s_ids, mask


def concat_embed(t_ids, c_ids, s_ids, P, glove):
    """ConcatInputer.get_embeddings (model/inputer/concat_inputer.py:92-114): sum of the three
    masked full-length look-ups (mask = id > -1, so the PAD(0) special ids past the prefix ARE
    looked up and added -- the operator's attention mask removes them later)."""
    if glove:
        e_t, _ = glove_project(t_ids, P["embedding_vocab_table.glove.embedding.weight"],
                               P["embedding_vocab_table.glove.linear.weight"],
                               P["embedding_vocab_table.glove.linear.bias"])
    else:
        e_t, _ = table_lookup(t_ids, P["embedding_vocab_table.glove.weight"])
    e_c, _ = table_lookup(c_ids, P["embedding_vocab_table.category.weight"])
    e_s, _ = table_lookup(s_ids, P["embedding_vocab_table.__cat_inputer_special_ids.weight"])
    return e_t + e_c + e_s


# --------------------------------------------------------------------------- a9 / a10
def dot_scores(user, items):
    """prepare_for_predictor + DotPredictor.predict (model/operators/base_operator.py:65-69,
    model/predictors/dot_predictor.py:7-10, model/legommender.py:268-283). user:[B,D] items:[B,C,D]"""
    return (user.unsqueeze(1) * items).sum(-1)


def ce_label0(scores):
    """nn.CrossEntropyLoss with labels == 0 (model/legommender.py:114-118,254,263)."""
    return F.cross_entropy(scores, torch.zeros(scores.shape[0], dtype=torch.long))


# --------------------------------------------------------------------------- a1 / a2 / a7
def _item_ids(cand, hist):
    """History pads are item 0 and ARE encoded by the reference (loader/resampler.py:222-223)."""
    B, C = cand.shape
    S = hist.shape[1]
    return torch.cat([cand.reshape(-1), hist.reshape(-1)]), B, C, S


def naml_forward(P: Dict[str, torch.Tensor], title_tok, cat, cand, hist, hist_len):
    """Legommender.forward for NAML/GloVe (model/legommender.py:219-263), eval-mode logits [B,C]."""
    ids, B, C, S = _item_ids(cand, hist)
    tok = title_tok[ids]
    emb, mask = glove_project(tok, P["embedding_vocab_table.glove.embedding.weight"],
                              P["embedding_vocab_table.glove.linear.weight"],
                              P["embedding_vocab_table.glove.linear.bias"])
    cat_emb = F.embedding(cat[ids], P["embedding_vocab_table.category.weight"]).unsqueeze(1)
    items = cnn_operator(emb, mask, cat_emb, P)                       # [B*(C+S), D]
    D = items.shape[-1]
    cand_v = items[: B * C].view(B, C, D)
    hist_v = items[B * C:].view(B, S, D)
    hmask = (torch.arange(S)[None, :] < hist_len[:, None]).long()     # __clicks_mask__
    user = additive_attention(hist_v, hmask,                           # AdaOperator (ada_operator.py:31-34)
                              P["user_op.additive_attention.encoder.0.weight"],
                              P["user_op.additive_attention.encoder.0.bias"],
                              P["user_op.additive_attention.encoder.2.weight"])
    return dot_scores(user, cand_v)


def nrms_forward(P, title_tok, title_len, cat, cand, hist, hist_len, heads=8, glove=False):
    """Legommender.forward for NRMS (item + user AttentionOperator), eval-mode logits [B,C]."""
    ids, B, C, S = _item_ids(cand, hist)
    t_ids, c_ids, s_ids, mask = concat_layout(title_tok[ids], title_len[ids], cat[ids], use_sep=True)
    x = concat_embed(t_ids, c_ids, s_ids, P, glove)
    items = attention_operator(x, mask, P, "item_op.", heads)
    D = items.shape[-1]
    cand_v = items[: B * C].view(B, C, D)
    hist_v = items[B * C:].view(B, S, D)
    hmask = (torch.arange(S)[None, :] < hist_len[:, None]).long()
    user = attention_operator(hist_v, hmask, P, "user_op.", heads)
    return dot_scores(user, cand_v)


# --------------------------------------------------------------------------- 8(f)-2: BERT news encoder
def bert_encoder(x, mask, P, prefix, n_layers, heads, eps=1e-12):
    """`BertModel(inputs_embeds=x, attention_mask=mask).last_hidden_state` with `word_embeddings = None`
    (reference call sites: model/operators/once_operator.py:156-170 `_forward`, bert_operator.py:16).

    The arithmetic lives in the third-party `transformers` package (reference pin `transformers~=4.47.1`,
    requirements.txt:10; the fixtures were generated with the version recorded in their meta).  Published algorithm
    (Devlin et al. 2019; modeling_bert.py BertEmbeddings / BertSelfAttention / BertSelfOutput / BertIntermediate /
    BertOutput): h0 = LayerNorm(x + type_emb[0] + pos_emb[:L]); per layer: scores = QK^T/sqrt(dh) + (1-mask)*finfo.min
    over keys, softmax, context; a = LayerNorm(dense(context) + h); h = LayerNorm(dense(gelu(dense(a))) + a).  x:[n,L,H]."""
    return bert_hidden_states(x, mask, P, prefix, n_layers, heads, eps)[-1]


def _bert_layer(h, ext, P, lp, heads, eps):
    n, L, H = h.shape
    dh = H // heads

    def heads_of(name):
        y = F.linear(h, P[lp + f"attention.self.{name}.weight"], P[lp + f"attention.self.{name}.bias"])
        a_key = lp + f"attention.self.{name}.lora_A.default.weight"
        if a_key in P:
            # LoRA (peft is a third-party dependency of the reference, requirements.txt `peft`, absent here -- published algorithm,
            # Hu et al. 2021 / peft.tuners.lora.Linear.forward: result = base(x) + lora_B(lora_A(dropout(x))) * (lora_alpha / r);
            # call site once_operator.py:137-151, bert_operator.py:26-28).  Dropout 0 in this restatement.
            y = y + F.linear(F.linear(h, P[a_key]), P[lp + f"attention.self.{name}.lora_B.default.weight"]) * float(P["__lora_scaling__"])
        return y.view(n, L, heads, dh).permute(0, 2, 1, 3)
    q, k, v = heads_of("query"), heads_of("key"), heads_of("value")
    probs = torch.softmax(q @ k.transpose(-1, -2) / math.sqrt(dh) + ext, dim=-1)
    ctx = (probs @ v).permute(0, 2, 1, 3).reshape(n, L, H)
    a = F.layer_norm(F.linear(ctx, P[lp + "attention.output.dense.weight"], P[lp + "attention.output.dense.bias"]) + h,
                     (H,), P[lp + "attention.output.LayerNorm.weight"], P[lp + "attention.output.LayerNorm.bias"], eps)
    i = F.gelu(F.linear(a, P[lp + "intermediate.dense.weight"], P[lp + "intermediate.dense.bias"]))
    return F.layer_norm(F.linear(i, P[lp + "output.dense.weight"], P[lp + "output.dense.bias"]) + a,
                        (H,), P[lp + "output.LayerNorm.weight"], P[lp + "output.LayerNorm.bias"], eps)


def bert_hidden_states(x, mask, P, prefix, n_layers, heads, eps=1e-12):
    """`BertModel(..., output_hidden_states=True).hidden_states`: [embeddings output, block 1 output, ..., block n output]
    -- entry k is what `OnceOperator.cache([k])` stores per item (once_operator.py:99-126, lm_layer_pager.py `combine`)."""
    n, L, H = x.shape
    h = x + P[prefix + "embeddings.token_type_embeddings.weight"][0] + P[prefix + "embeddings.position_embeddings.weight"][:L]
    h = F.layer_norm(h, (H,), P[prefix + "embeddings.LayerNorm.weight"], P[prefix + "embeddings.LayerNorm.bias"], eps)
    ext = (1.0 - mask.to(x.dtype))[:, None, None, :] * torch.finfo(x.dtype).min
    states = [h]
    for l in range(n_layers):
        h = _bert_layer(h, ext, P, f"{prefix}encoder.layer.{l}.", heads, eps)
        states.append(h)
    return states


def bert_operator_cached(hidden, mask, P, prefix, n_layers, heads, eps=1e-12):
    """OnceOperator.forward, `tune_from = k > 0` branch (once_operator.py:182-193, bert_operator.py:30-45): the cached
    layer-k states of the batch's items run through the blocks KEPT after `encoder.layer[k + 1:]` (renumbered from 0 in
    the state_dict; block k itself is skipped -- the reference's off-by-one), then Linear and the additive pool."""
    ext = (1.0 - mask.to(hidden.dtype))[:, None, None, :] * torch.finfo(hidden.dtype).min
    h = hidden
    for l in range(n_layers):
        h = _bert_layer(h, ext, P, f"{prefix}transformer.encoder.layer.{l}.", heads, eps)
    y = F.linear(h, P[prefix + "linear.weight"], P[prefix + "linear.bias"])
    return additive_attention(y, mask, P[prefix + "additive_attention.encoder.0.weight"],
                              P[prefix + "additive_attention.encoder.0.bias"],
                              P[prefix + "additive_attention.encoder.2.weight"])


def bert_layer_cache(P_ckpt, word_table, cat_table, title_tok, title_len, cat, layer, n_layers, heads, eps=1e-12):
    """The per-item cache of `OnceOperator.cache([layer])`: hidden_states[layer] of the CHECKPOINT transformer (all blocks,
    before slicing) for every item, and the ConcatInputer mask.  P_ckpt: BertModel state_dict keys (no prefix)."""
    t_ids, c_ids, _, mask = concat_layout(title_tok, title_len, cat, use_sep=False)
    e_t, _ = table_lookup(t_ids, word_table)
    e_c, _ = table_lookup(c_ids, cat_table)
    return bert_hidden_states(e_t + e_c, mask, P_ckpt, "", n_layers, heads, eps)[layer], mask


def bert_naml_cached_forward(P, hidden_cache, mask_cache, cand, hist, hist_len, n_layers, heads, eps=1e-12):
    """Legommender.forward with `Env.lm_cache` (legommender.py:166-189): item ids index the layer cache directly."""
    ids, B, C, S = _item_ids(cand, hist)
    items = bert_operator_cached(hidden_cache[ids], mask_cache[ids], P, "item_op.", n_layers, heads, eps)
    D = items.shape[-1]
    cand_v = items[: B * C].view(B, C, D)
    hist_v = items[B * C:].view(B, S, D)
    hmask = (torch.arange(S)[None, :] < hist_len[:, None]).long()
    user = additive_attention(hist_v, hmask, P["user_op.additive_attention.encoder.0.weight"],
                              P["user_op.additive_attention.encoder.0.bias"],
                              P["user_op.additive_attention.encoder.2.weight"])
    return dot_scores(user, cand_v)


def bert_operator(x, mask, P, prefix, n_layers, heads, eps=1e-12):
    """OnceOperator.forward, `tune_from` falsy branch (once_operator.py:173-193): transformer -> Linear -> additive pool.
    `n_layers` = layers KEPT: with the yaml default tune_from = 0 the constructor still slices `encoder.layer[1:]`
    (once_operator.py:128-134, bert_operator.py:23-24) and forward runs the remaining layers on the input embeddings."""
    h = bert_encoder(x, mask, P, prefix + "transformer.", n_layers, heads, eps)
    y = F.linear(h, P[prefix + "linear.weight"], P[prefix + "linear.bias"])
    return additive_attention(y, mask, P[prefix + "additive_attention.encoder.0.weight"],
                              P[prefix + "additive_attention.encoder.0.bias"],
                              P[prefix + "additive_attention.encoder.2.weight"])


def bert_naml_forward(P, title_tok, title_len, cat, cand, hist, hist_len, n_layers, heads, eps=1e-12):
    """Legommender.forward for config/model/bert-naml.yaml (item = BertBase over ConcatInputer without CLS/SEP,
    user = Ada, predictor = Dot), eval-mode logits [B,C]."""
    ids, B, C, S = _item_ids(cand, hist)
    t_ids, c_ids, _, mask = concat_layout(title_tok[ids], title_len[ids], cat[ids], use_sep=False)
    e_t, _ = table_lookup(t_ids, P["embedding_vocab_table.glove.weight"])
    e_c, _ = table_lookup(c_ids, P["embedding_vocab_table.category.weight"])
    items = bert_operator(e_t + e_c, mask, P, "item_op.", n_layers, heads, eps)
    D = items.shape[-1]
    cand_v = items[: B * C].view(B, C, D)
    hist_v = items[B * C:].view(B, S, D)
    hmask = (torch.arange(S)[None, :] < hist_len[:, None]).long()
    user = additive_attention(hist_v, hmask, P["user_op.additive_attention.encoder.0.weight"],
                              P["user_op.additive_attention.encoder.0.bias"],
                              P["user_op.additive_attention.encoder.2.weight"])
    return dot_scores(user, cand_v)


def eval_scores(kind, P, title_tok, title_len, cat, user_hist, user_hist_len, rows_user, rows_item, heads=8, glove=True):
    """The reference's fast-eval path: all-item cache (loader/cacher/item_cacher.py:51-97), all-user cache from
    `item_repr[history]` (loader/cacher/user_cacher.py:63-97, model/legommender.py:153-157,202-214), then
    score = <user_repr[u], item_repr[i]> per evaluation row (legommender.py:282)."""
    n_items = title_tok.shape[0]
    S = user_hist.shape[1]
    if kind == "naml":
        emb, mask = glove_project(title_tok, P["embedding_vocab_table.glove.embedding.weight"],
                                  P["embedding_vocab_table.glove.linear.weight"], P["embedding_vocab_table.glove.linear.bias"])
        cat_emb = F.embedding(cat, P["embedding_vocab_table.category.weight"]).unsqueeze(1)
        item_repr = cnn_operator(emb, mask, cat_emb, P)
    else:
        t_ids, c_ids, s_ids, mask = concat_layout(title_tok, title_len, cat, use_sep=True)
        item_repr = attention_operator(concat_embed(t_ids, c_ids, s_ids, P, glove), mask, P, "item_op.", heads)
    hmask = (torch.arange(S)[None, :] < user_hist_len[:, None]).long()
    clicks = item_repr[user_hist]                                   # pads are item 0, masked below
    if kind == "naml":
        user_repr = additive_attention(clicks, hmask, P["user_op.additive_attention.encoder.0.weight"],
                                       P["user_op.additive_attention.encoder.0.bias"],
                                       P["user_op.additive_attention.encoder.2.weight"])
    else:
        user_repr = attention_operator(clicks, hmask, P, "user_op.", heads)
    return (user_repr[rows_user] * item_repr[rows_item]).sum(-1), item_repr, user_repr


def loss_and_grads(kind, P_np, tables, cand, hist, hist_len, heads=8, glove=True, frozen=(), bert_layers=0, bert_eps=1e-12,
                   layer_cache=None, train_table=False):
    """Logits, loss and d(loss)/d(param) for every trainable tensor (dropout 0).  numpy in/out.
    `train_table`: the pre-trained token table is un-frozen (`load_pretrained_embedding(..., frozen=False)`,
    loader/embedding_hub.py:171,262 -- its weight gets requires_grad and the dense gradient of the look-up)."""
    P = {}
    for k, v in P_np.items():
        if k.startswith("__"):                                  # non-tensor settings riding along (e.g. __lora_scaling__)
            P[k] = v
            continue
        t = _t(v).clone()
        if k not in frozen and t.dtype == torch.float32 and (train_table or not k.endswith("glove.embedding.weight")):
            t.requires_grad_(True)
        P[k] = t
    tt, tl, ct = _t(tables["title_tok"]), _t(tables["title_len"]), _t(tables["cat"])
    c, h, hl = _t(cand), _t(hist), _t(hist_len)
    if kind == "naml":
        logits = naml_forward(P, tt, ct, c, h, hl)
    elif kind == "bert_naml" and layer_cache is not None:
        logits = bert_naml_cached_forward(P, _t(layer_cache[0]), _t(layer_cache[1]), c, h, hl, bert_layers, heads, bert_eps)
    elif kind == "bert_naml":
        logits = bert_naml_forward(P, tt, tl, ct, c, h, hl, bert_layers, heads, bert_eps)
    else:
        logits = nrms_forward(P, tt, tl, ct, c, h, hl, heads=heads, glove=glove)
    loss = ce_label0(logits)
    names = [k for k, v in P.items() if isinstance(v, torch.Tensor) and v.requires_grad]
    grads = torch.autograd.grad(loss, [P[k] for k in names], allow_unused=True)
    g = {k: (gv.numpy() if gv is not None else np.zeros_like(P_np[k])) for k, gv in zip(names, grads)}
    return logits.detach().numpy(), float(loss.detach()), g


def naml_train_step_cpu(P, opt, title_tok, cat, cand, hist, hist_len, p_drop=0.1):
    """One TRAINING step of the reference's CPU path (`--cuda -1`), dense layout, dropout on:
    forward (model/legommender.py:219-263 with Dropout at loader/embedding_hub.py:96 and
    model/operators/cnn_operator.py:57), loss.backward(), Adam step (trainer.py:193-203).
    `P` holds torch tensors (requires_grad on the trainable ones), `opt` a torch.optim.Adam over them.
    Used as the `cpu_baseline` ("port") of bench.py and by tests; returns the loss value."""
    ids, B, C, S = _item_ids(cand, hist)
    tok = title_tok[ids]
    mask = (tok > UNSET).long()
    emb = F.linear(F.embedding(tok * mask, P["embedding_vocab_table.glove.embedding.weight"]),
                   P["embedding_vocab_table.glove.linear.weight"], P["embedding_vocab_table.glove.linear.bias"])
    emb = F.dropout(emb, p_drop, training=True) * mask.unsqueeze(-1).to(emb.dtype)
    cat_emb = F.embedding(cat[ids], P["embedding_vocab_table.category.weight"]).unsqueeze(1)
    y = F.conv1d(emb.permute(0, 2, 1), P["item_op.cnn.weight"], P["item_op.cnn.bias"], padding="same")
    y = F.dropout(torch.relu(y.permute(0, 2, 1)) * mask.unsqueeze(-1).to(emb.dtype), p_drop, training=True)
    c = F.linear(cat_emb, P["item_op.linear.weight"], P["item_op.linear.bias"])
    out = torch.cat([y, c], dim=1)
    m2 = torch.cat([mask, torch.ones_like(mask[:, :1])], dim=1)
    items = additive_attention(out, m2, P["item_op.additive_attention.encoder.0.weight"],
                               P["item_op.additive_attention.encoder.0.bias"],
                               P["item_op.additive_attention.encoder.2.weight"])
    D = items.shape[-1]
    hmask = (torch.arange(S)[None, :] < hist_len[:, None]).long()
    user = additive_attention(items[B * C:].view(B, S, D), hmask,
                              P["user_op.additive_attention.encoder.0.weight"],
                              P["user_op.additive_attention.encoder.0.bias"],
                              P["user_op.additive_attention.encoder.2.weight"])
    loss = ce_label0(dot_scores(user, items[: B * C].view(B, C, D)))
    opt.zero_grad()
    loss.backward()
    opt.step()
    return float(loss.detach())


# --------------------------------------------------------------------------- a13
def linear_schedule_factor(step, total, warmup=0):
    """transformers.get_linear_schedule_with_warmup lambda (base_lego.py:211-223)."""
    if step < warmup:
        return float(step) / float(max(1, warmup))
    return max(0.0, float(total - step) / float(max(1, total - warmup)))


def adam_step(p, g, m, v, step, lr, b1=0.9, b2=0.999, eps=1e-8):
    """torch.optim.Adam defaults, no weight decay, no amsgrad (base_lego.py:201-204). step is 1-based.
    fp32 arrays; returns (p, m, v)."""
    p, g, m, v = (np.asarray(a, dtype=np.float32) for a in (p, g, m, v))
    m = (b1 * m + (1 - b1) * g).astype(np.float32)
    v = (b2 * v + (1 - b2) * g * g).astype(np.float32)
    bc1 = 1.0 - b1 ** step
    bc2 = 1.0 - b2 ** step
    denom = (np.sqrt(v) / np.float32(math.sqrt(bc2)) + np.float32(eps)).astype(np.float32)
    p = (p - np.float32(lr / bc1) * (m / denom)).astype(np.float32)
    return p, m, v


# --------------------------------------------------------------------------- a11
def sample_negatives_semantics(cand_row, true_negs, item_size, K=4):
    """Property checker for the negative sampler (loader/resampler.py:159-171): the first
    min(K,len(true_negs)) negatives are distinct draws from the user's true-negative list, the
    remainder are uniform item ids in [0,item_size-1] (collisions allowed).  Returns bool."""
    negs = list(cand_row[1:])
    if len(negs) != K:
        return False
    n_true = min(K, len(true_negs))
    pool = list(true_negs)
    for x in negs[:n_true]:
        if x not in pool:
            return False
        pool.remove(x)
    return all(0 <= x < item_size for x in negs[n_true:])


# --------------------------------------------------------------------------- a14 (metrics)
def _auc(labels, scores):
    """sklearn.roc_auc_score restated: Mann-Whitney U with average ranks for ties."""
    order = np.argsort(scores, kind="mergesort")
    s = scores[order]
    ranks = np.empty(len(s), dtype=np.float64)
    i = 0
    while i < len(s):
        j = i
        while j + 1 < len(s) and s[j + 1] == s[i]:
            j += 1
        ranks[i:j + 1] = 0.5 * (i + j) + 1.0
        i = j + 1
    r = np.empty_like(ranks)
    r[order] = ranks
    pos = labels == 1
    n_pos, n_neg = pos.sum(), (~pos).sum()
    return (r[pos].sum() - n_pos * (n_pos + 1) / 2.0) / (n_pos * n_neg)


def _dcg(rel, scores, k):
    """sklearn.metrics.ndcg_score(ignore_ties=False) DCG with tie averaging, truncated at k."""
    disc = 1.0 / np.log2(np.arange(len(rel)) + 2.0)
    disc[k:] = 0.0
    _, inv, counts = np.unique(-scores, return_inverse=True, return_counts=True)
    ranked = np.zeros(len(counts))
    np.add.at(ranked, inv, rel)
    ranked /= counts
    groups = np.cumsum(counts) - 1
    dsum = np.empty(len(counts))
    cs = np.cumsum(disc)
    dsum[0] = cs[groups[0]]
    dsum[1:] = np.diff(cs[groups])
    return float((ranked * dsum).sum())


def _ndcg(labels, scores, k):
    ideal = _dcg(labels.astype(np.float64), labels.astype(np.float64), k)
    return _dcg(labels.astype(np.float64), scores.astype(np.float64), k) / ideal if ideal > 0 else 0.0


def _mrr(labels, scores):
    """The reference's non-standard MRR (utils/metrics.py:144-160): mean over ALL positives of
    1/rank, divided by the number of positives."""
    order = np.argsort(-scores, kind="stable")
    y = labels[order]
    rr = y / (np.arange(len(y)) + 1.0)
    return float(rr.sum() / y.sum())


def _ranked_labels(labels, scores):
    """labels in the order of Python's `sorted(range(n), key=scores.__getitem__, reverse=True)` (ties keep row order),
    the ranking MRR / MRR0 / HitRatio / Recall share (utils/metrics.py:131,153,195,212)"""
    order = sorted(range(len(scores)), key=lambda i: scores[i], reverse=True)
    return [int(labels[i]) for i in order]


def _mrr0(labels, scores):
    """utils/metrics.py:125-140"""
    for rank, y in enumerate(_ranked_labels(labels, scores), start=1):
        if y == 1:
            return 1.0 / rank
    return 0.0


def _hit_ratio(labels, scores, k):
    """utils/metrics.py:183-197"""
    return float(1 in _ranked_labels(labels, scores)[:k])


def _recall(labels, scores, k):
    """utils/metrics.py:199-214"""
    y = _ranked_labels(labels, scores)
    return sum(y[:k]) * 1.0 / sum(y)


def _lrap(labels, scores):
    """sklearn.label_ranking_average_precision_score of ONE sample (utils/metrics.py:108-117): mean over the relevant
    labels of (#relevant ranked at or above it) / (#labels ranked at or above it), ties counted with `>=`;
    1.0 when no label or every label is relevant."""
    rel = [i for i in range(len(labels)) if labels[i] == 1]
    if len(rel) == 0 or len(rel) == len(labels):
        return 1.0
    acc = 0.0
    for i in rel:
        at_or_above = [j for j in range(len(scores)) if scores[j] >= scores[i]]
        acc += sum(1 for j in at_or_above if labels[j] == 1) / len(at_or_above)
    return acc / len(rel)


def pointwise_metric(name, scores, labels):
    """AUC / LogLoss / F1@t over all rows (utils/metrics.py:66-96,162-181; sklearn roc_auc_score, log_loss, f1_score)"""
    scores = np.asarray(scores, dtype=np.float64)
    labels = np.asarray(labels)
    base, _, arg = name.partition("@")
    if base == "AUC":
        return float(_auc(labels, scores))
    if base == "LogLoss":
        eps = np.finfo(np.float64).eps
        p = np.clip(scores, eps, 1 - eps)
        return float(np.mean([-np.log(pi) if y == 1 else -np.log1p(-pi) for pi, y in zip(p, labels)]))
    if base == "F1":
        t = float(arg) if arg else 0.5
        pred = [int(v >= t) for v in scores]
        tp = sum(1 for q, y in zip(pred, labels) if q == 1 and y == 1)
        fp = sum(1 for q, y in zip(pred, labels) if q == 1 and y != 1)
        fn = sum(1 for q, y in zip(pred, labels) if q == 0 and y == 1)
        return 2.0 * tp / (2.0 * tp + fp + fn) if tp + fp + fn else 0.0
    raise ValueError(name)


def grouped_metrics(scores, labels, groups, names=("GAUC", "MRR", "NDCG@1", "NDCG@5", "NDCG@10")):
    """MetricPool.calculate (utils/metrics.py:313-369): per-`group` metric, fp32 mean over groups; point-wise metrics
    (AUC, LogLoss, F1) over all rows."""
    scores = np.asarray(scores, dtype=np.float64)
    labels = np.asarray(labels)
    groups = np.asarray(groups)
    out = {}
    uniq = np.unique(groups)
    for name in names:
        base, _, arg = name.partition("@")
        if base in ("AUC", "LogLoss", "F1"):
            out[name] = pointwise_metric(name, scores, labels)
            continue
        vals = []
        for g in uniq:
            sel = groups == g
            l, s = labels[sel], scores[sel]
            if name == "GAUC":
                vals.append(_auc(l, s))
            elif name == "MRR":
                vals.append(_mrr(l, s))
            elif name == "MRR0":
                vals.append(_mrr0(l, s))
            elif name == "LRAP":
                vals.append(_lrap(l, s))
            elif base == "NDCG":
                vals.append(_ndcg(l, s, int(arg)))
            elif base == "HitRatio":
                vals.append(_hit_ratio(l, s, int(arg)))
            elif base == "Recall":
                vals.append(_recall(l, s, int(arg)))
            else:
                raise ValueError(name)
        out[name] = float(np.asarray(vals, dtype=np.float32).mean(dtype=np.float32))
    return out
