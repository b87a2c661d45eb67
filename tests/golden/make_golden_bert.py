#!/usr/bin/env python3
"""Golden fixture for SURVEY.md section 8(f)-2: the reference's BERT news encoder (`BertBaseOperator`, config
`bert-naml.yaml`: item = BertBase, user = Ada, predictor = Dot) on a tiny RANDOM-INIT BertConfig (no pretrained
weights exist offline).  Container-only, like make_golden.py: imports the reference with in-memory stubs
(`unitok`, `pigmento`, plus `peft` and `utils.config_init`, which this operator pulls in) and stores data only.

    python tests/golden/make_golden_bert.py       # rewrites tests/golden/bert_naml_small.npz
    python tests/golden/make_golden_bert.py tune1 # rewrites tests/golden/bert_naml_tune1.npz (tune_from = 1)
"""
from __future__ import annotations

import json
import os
import random
import shutil
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as MG  # noqa: E402

TMP = "/tmp/_golden_bert"


def install_bert_stubs(bert_dir):
    MG.install_stubs()
    peft = types.ModuleType("peft")
    peft.get_peft_model = lambda m, c: m
    peft.LoraConfig = lambda **k: None
    peft.PeftConfig = object
    sys.modules["peft"] = peft
    ci = types.ModuleType("utils.config_init")

    class ModelInit:
        @classmethod
        def get(cls, name):
            return bert_dir
    ci.ModelInit = ModelInit
    sys.modules["utils.config_init"] = ci


def main(tune_from=0, name="bert_naml_small"):
    from transformers import BertConfig, BertModel
    seed, D, H, V, n_items, n_users, B = 31 + tune_from, 32, 64, 300, 60, 30, 6
    torch.manual_seed(seed); random.seed(seed); np.random.seed(seed)
    shutil.rmtree(TMP, ignore_errors=True)
    os.makedirs(TMP)
    cfg = BertConfig(vocab_size=V, hidden_size=H, num_hidden_layers=3, num_attention_heads=4, intermediate_size=96,
                     max_position_embeddings=64, hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0)
    BertModel(cfg).save_pretrained(TMP)
    install_bert_stubs(TMP)
    os.chdir(TMP)                              # the operator writes cache/<data>/<name>/layer_0.npy relative to cwd
    from loader.env import Env
    Env.device = torch.device("cpu")
    Env.ph = types.SimpleNamespace(data_name="golden")
    from loader.column_map import ColumnMap
    from loader.embedding_hub import EmbeddingHub
    from model.lego_config import LegoConfig
    from model.legommender import Legommender
    from model.operators.ada_operator import AdaOperator
    from model.operators.bert_operator import BertBaseOperator
    from model.predictors.dot_predictor import DotPredictor
    from loader.resampler import Resampler
    del sys.modules["peft"]                    # only the reference's import statements need it; transformers must not see it

    w = MG.make_world(seed, V, n_items, n_users, n_rows=4 * B)
    tok_v = MG.SizedVocab("glove", w["V"])       # plays the word-piece vocabulary; its table has the BERT hidden size
    cat_v = MG.SizedVocab("category", w["n_cat"])
    item_v = MG.SizedVocab("item_id", w["n_items"])
    user_v = MG.SizedVocab("user_id", w["n_users"])
    item_rows = [{"item_id": i, "title@glove": w["title_tok"][i, : w["title_len"][i]].tolist(),
                  "category": int(w["cat"][i])} for i in range(w["n_items"])]
    item_ut = MG.FakeUT(item_rows, [MG.Feat("item_id", item_v), MG.Feat("title@glove", tok_v, w["T"]),
                                    MG.Feat("category", cat_v)], "item_id")
    user_rows = [{"user_id": u, "history": list(w["hist"][u]), "neg": list(w["neg"][u])} for u in range(w["n_users"])]
    user_ut = MG.FakeUT(user_rows, [MG.Feat("user_id", user_v), MG.Feat("history", item_v, w["S"]),
                                    MG.Feat("neg", item_v, 100)], "user_id")
    inter_rows = [{"index": r, "user_id": int(w["row_user"][r]), "item_id": int(w["row_item"][r]), "click": 1,
                   "history": list(w["hist"][w["row_user"][r]]), "neg": list(w["neg"][w["row_user"][r]])}
                  for r in range(len(w["row_user"]))]
    inter_ut = MG.FakeUT(inter_rows, [MG.Feat("index", MG.SizedVocab("index", len(inter_rows))),
                                      MG.Feat("user_id", user_v), MG.Feat("item_id", item_v),
                                      MG.Feat("click", MG.SizedVocab("click", 2)),
                                      MG.Feat("history", item_v, w["S"]), MG.Feat("neg", item_v, 100)], "index")
    # config/model/bert-naml.yaml: item_hidden_size = transformer width, tune_from 0, no LoRA, no CLS/SEP tokens
    lc = LegoConfig(hidden_size=D, item_hidden_size=H, neg_count=4, item_page_size=0,
                    user_config={"inputer_config": {"use_cls_token": False, "use_sep_token": False}},
                    item_config={"tune_from": tune_from, "use_lora": False, "lora_r": None, "lora_alpha": None,
                                 "inputer_config": {"use_cls_token": False, "use_sep_token": False}})
    lc.set_component_classes(BertBaseOperator, AdaOperator, DotPredictor)
    lc.set_item_ut(item_ut, ["title@glove", "category"])
    lc.set_user_ut(user_ut, ["history"])
    lc.set_column_map(ColumnMap(item_col="item_id", user_col="user_id", history_col="history",
                                neg_col="neg", label_col="click", group_col="user_id"))
    eh = EmbeddingHub(embedding_dim=H, transformation="auto", transformation_dropout=0.0)
    path = os.path.join(TMP, "wordpiece.npy")
    np.save(path, MG.glove_table(seed + 1, w["V"], E0=H))
    eh.load_pretrained_embedding(path, vocab_name="glove", frozen=True)
    eh.register_ut(item_ut, ["title@glove", "category"])
    lc.set_embedding_hub(eh)
    lc.build_components()
    lc.register_inputer_vocabs()
    model = Legommender(lc)
    kept = len(model.item_op.transformer.encoder.layer)
    assert kept == cfg.num_hidden_layers - 1 - tune_from   # tune_from = 0 still slices layer[1:] (once_operator.py:128-134)
    assert Env.lm_cache == bool(tune_from)
    with torch.no_grad():                             # non-trivial biases / LayerNorm gains so they are pinned too
        for n, p_ in model.named_parameters():
            if p_.requires_grad and (n.endswith("bias") or "LayerNorm.weight" in n):
                p_.add_(torch.randn_like(p_) * 0.05)
    resampler = Resampler(lc)
    Env.train(); model.train()
    batch = MG.reference_batch(resampler, inter_ut, list(range(B)))
    if tune_from:                                     # Env.lm_cache: the resampler hands over bare ids (resampler.py:181-188,244-246)
        cand = batch["item_id"].numpy().copy()
        hist = batch["history"].numpy().copy()
        hist_len = batch["__clicks_mask__"].sum(-1).numpy().copy()
    else:
        cand, hist, hist_len = MG.flat_batch("concat", batch, w)
    assert (cand[:, 0] == w["row_item"][:B]).all()
    batch2 = MG.clone_batch(batch)
    model.zero_grad()
    loss = model(batch=batch)
    loss.backward()
    grads = {"grad::" + n: p_.grad.detach().numpy().copy()
             for n, p_ in model.named_parameters() if p_.requires_grad and p_.grad is not None}
    Env.test(); model.eval()
    with torch.no_grad():
        logits = model(batch=batch2).numpy().copy()
    out = {}
    out.update(MG.state_np(model))
    out.update(grads)
    out.update(MG.world_np(w))
    out.update({"cand": cand, "hist": hist, "hist_len": hist_len, "logits": logits, "loss": np.float32(loss.item())})
    if tune_from:
        # the layer cache the operator trained on (once_operator.py:99-126) and the checkpoint it was computed from
        out["cache::hidden"] = model.item_op.hidden_weights.numpy().copy()
        out["cache::mask"] = model.item_op.attention_mask.numpy().copy()
        for k_, v_ in BertModel.from_pretrained(TMP).state_dict().items():
            out["ckpt::" + k_] = v_.numpy().copy()
    bert_cfg = dict(hidden_size=H, num_hidden_layers=cfg.num_hidden_layers, num_attention_heads=cfg.num_attention_heads,
                    intermediate_size=cfg.intermediate_size, max_position_embeddings=cfg.max_position_embeddings,
                    layer_norm_eps=cfg.layer_norm_eps, hidden_act=cfg.hidden_act, vocab_size=V)
    meta = dict(kind="bert_naml", embed="wordpiece", D=D, item_hidden=H, V=V, n_items=n_items, n_users=n_users, B=B, seed=seed,
                heads=cfg.num_attention_heads, tune_from=tune_from, layers_kept=kept, bert=bert_cfg, torch=torch.__version__,
                transformers=__import__("transformers").__version__,
                note="random-init BertConfig (no pretrained weights offline); the frozen word-piece table is stored")
    out["meta"] = np.frombuffer(json.dumps(meta).encode(), dtype=np.uint8)
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)
    print(name, ": loss", float(loss.detach()), "logits[0]", logits[0], "grads", len(grads))
    shutil.rmtree(TMP, ignore_errors=True)


if __name__ == "__main__":
    if sys.argv[1:] == ["tune1"]:
        main(1, "bert_naml_tune1")        # cached-layer mode: hidden states of layer 1 cached, blocks [2:] train
    else:
        main()
