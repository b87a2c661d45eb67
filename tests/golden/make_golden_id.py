#!/usr/bin/env python3
"""Golden fixture of the ID-based branch of `Legommender.forward` (model/legommender.py:237-248, config/model/naml_id.yaml:
`use_item_content: false`): candidates and clicked items are embedded by their ITEM ID through the embedding hub -- no item
operator -- and the user operator (AdaOperator) pools the clicked items' id embeddings.  Runs ONLY in the build container
(imports /root/reference through make_golden's stubs); stores data only: tests/golden/naml_id_d64.npz.

    python tests/golden/make_golden_id.py"""
from __future__ import annotations

import json
import os
import random
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as MG                                              # noqa: E402


def build_id_model(world, D):
    """Manager.__init__ order for a model without an item operator (loader/manager.py:281-283,321-324)"""
    from loader.env import Env
    Env.device = torch.device("cpu")
    from loader.column_map import ColumnMap
    from loader.embedding_hub import EmbeddingHub
    from loader.resampler import Resampler
    from model.lego_config import LegoConfig
    from model.legommender import Legommender
    from model.operators.ada_operator import AdaOperator
    from model.predictors.dot_predictor import DotPredictor
    w = world
    glove_v, cat_v = MG.SizedVocab("glove", w["V"]), MG.SizedVocab("category", w["n_cat"])
    item_v, user_v = MG.SizedVocab("item_id", w["n_items"]), MG.SizedVocab("user_id", w["n_users"])
    item_rows = [{"item_id": i, "title@glove": w["title_tok"][i, : w["title_len"][i]].tolist(), "category": int(w["cat"][i])}
                 for i in range(w["n_items"])]
    item_ut = MG.FakeUT(item_rows, [MG.Feat("item_id", item_v), MG.Feat("title@glove", glove_v, w["T"]), MG.Feat("category", cat_v)], "item_id")
    user_rows = [{"user_id": u, "history": list(w["hist"][u]), "neg": list(w["neg"][u])} for u in range(w["n_users"])]
    user_ut = MG.FakeUT(user_rows, [MG.Feat("user_id", user_v), MG.Feat("history", item_v, w["S"]), MG.Feat("neg", item_v, 100)], "user_id")
    inter_rows = [{"index": r, "user_id": int(w["row_user"][r]), "item_id": int(w["row_item"][r]), "click": 1,
                   "history": list(w["hist"][w["row_user"][r]]), "neg": list(w["neg"][w["row_user"][r]])} for r in range(len(w["row_user"]))]
    inter_ut = MG.FakeUT(inter_rows, [MG.Feat("index", MG.SizedVocab("index", len(inter_rows))), MG.Feat("user_id", user_v),
                                      MG.Feat("item_id", item_v), MG.Feat("click", MG.SizedVocab("click", 2)),
                                      MG.Feat("history", item_v, w["S"]), MG.Feat("neg", item_v, 100)], "index")
    lc = LegoConfig(hidden_size=D, item_hidden_size=D, neg_count=4, use_item_content=False,
                    user_config={"inputer_config": {"use_cls_token": False, "use_sep_token": False}})
    lc.set_component_classes(None, AdaOperator, DotPredictor)
    lc.set_item_ut(item_ut, ["title@glove", "category"])
    lc.set_user_ut(user_ut, ["history"])
    lc.set_column_map(ColumnMap(item_col="item_id", user_col="user_id", history_col="history", neg_col="neg", label_col="click", group_col="user_id"))
    eh = EmbeddingHub(embedding_dim=D, transformation="auto", transformation_dropout=0.0)
    eh.register_vocab(item_v)                                         # manager.py:323-324
    lc.set_embedding_hub(eh)
    lc.build_components()
    lc.register_inputer_vocabs()
    return Legommender(lc), Resampler(lc), inter_ut


def main():
    assert os.path.isdir(MG.REF), "the reference is only present in the build container"
    MG.install_stubs()
    torch.set_num_threads(4)
    from loader.env import Env
    name, D, V, n_items, n_users, B, seed = "naml_id_d64", 64, 500, 120, 40, 8, 2031
    torch.manual_seed(seed); random.seed(seed); np.random.seed(seed)
    world = MG.make_world(seed, V, n_items, n_users, n_rows=4 * B)
    model, resampler, inter_ut = build_id_model(world, D)
    with torch.no_grad():
        for n, p_ in model.named_parameters():
            if p_.requires_grad and n.endswith("bias"):
                p_.add_(torch.randn_like(p_) * 0.05)
    Env.train(); model.train()
    batch = MG.reference_batch(resampler, inter_ut, list(range(B)))
    cand = batch["item_id"].numpy().astype(np.int64)                  # ID mode: the resampler ships ids, not stacked content
    mask = batch["__clicks_mask__"].numpy()
    hb = batch["history"]                                             # ID mode: the user inputer's nested sample (resampler.py:222-226)
    assert set(hb) == {"input_ids", "attention_mask"} and set(hb["input_ids"]) == {"history"}, hb
    hist = hb["input_ids"]["history"].numpy().astype(np.int64)
    assert hist.shape == mask.shape and ((hist >= 0) == (mask == 1)).all()         # SimpleInputer pads with -1
    hist = hist * mask
    assert cand.shape == (B, 5) and (cand[:, 0] == world["row_item"][:B]).all()
    batch2 = MG.clone_batch(batch)
    model.zero_grad()
    loss = model(batch=batch)
    loss.backward()
    grads = {"grad::" + n: p_.grad.detach().numpy().copy() for n, p_ in model.named_parameters() if p_.requires_grad and p_.grad is not None}
    Env.test(); model.eval()
    with torch.no_grad():
        logits = model(batch=batch2).numpy().copy()
    Env.train()
    out = {}
    out.update(MG.state_np(model)); out.update(grads); out.update(MG.world_np(world))
    out.update({"cand": cand, "hist": hist, "hist_len": mask.sum(1), "logits": logits, "loss": np.float32(loss.item())})
    meta = dict(kind="naml_id", embed="id", D=D, V=V, n_items=n_items, n_users=n_users, B=B, seed=seed, heads=0, torch=torch.__version__,
                note="use_item_content=False: item-id embeddings (embedding_vocab_table.item_id.weight) for candidates and history")
    out["meta"] = np.frombuffer(json.dumps(meta).encode(), dtype=np.uint8)
    np.savez_compressed(os.path.join(MG.OUT, name + ".npz"), **out)
    print(name, "loss", float(loss), "logits[0]", logits[0], "params", sorted(k for k in out if k.startswith("param::")))


if __name__ == "__main__":
    main()
