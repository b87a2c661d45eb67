"""The reference's plug-in surface on the MI355X path: operators / predictors discovered through ClassHub,
built by LegoConfig exactly as `Manager.__init__` does (loader/manager.py:139-153,294-326), loaded with the
reference's own state_dict (golden fixtures) and driven through `Legommender.forward` -- both routes
(per-operator plug-in route and the fused engine route) must reproduce the reference's logits, loss and
gradients."""
import numpy as np
import pytest
import torch

from tests.golden_util import load_model_fixture


def _build(name, dev, frozen=True):
    from legommenders_amd.engine import ItemTables
    from legommenders_amd.loader.class_hub import ClassHub
    from legommenders_amd.loader.column_map import ColumnMap
    from legommenders_amd.loader.embedding_hub import EmbeddingHub
    from legommenders_amd.loader.env import Env
    from legommenders_amd.loader.tables import Feature, Table, Vocab
    from legommenders_amd.model.lego_config import LegoConfig
    from legommenders_amd.model.legommender import Legommender

    meta, P, G, tables, batch, logits, loss = load_model_fixture(name)
    Env.set_device(dev)
    D, V = meta["D"], meta["V"]
    n_items = tables["title_tok"].shape[0]
    glove_v, cat_v, item_v = Vocab("glove", V), Vocab("category", 18), Vocab("item_id", n_items)
    user_v = Vocab("user_id", tables["user_hist"].shape[0])
    item_ut = Table([Feature("item_id", item_v), Feature("title@glove", glove_v, 30), Feature("category", cat_v)],
                    {"item_id": np.arange(n_items), "title@glove": (tables["title_tok"], tables["title_len"]),
                     "category": tables["cat"]}, "item_id")
    user_ut = Table([Feature("user_id", user_v), Feature("history", item_v, 50)],
                    {"user_id": np.arange(user_v.size), "history": (tables["user_hist"], tables["user_hist_len"])}, "user_id")
    ops, preds = ClassHub.operators(), ClassHub.predictors()
    if meta["kind"] == "naml":
        lc = LegoConfig(hidden_size=D, item_hidden_size=D, neg_count=4,
                        user_config={"inputer_config": {"use_cls_token": False, "use_sep_token": False}},
                        item_config={"dropout": 0.0, "kernel_size": 3})
        lc.set_component_classes(ops["CNN"], ops["Ada"], preds["Dot"])
    else:
        lc = LegoConfig(hidden_size=D, item_hidden_size=D, neg_count=4,
                        item_config={"num_attention_heads": meta["heads"], "attention_dropout": 0.0,
                                     "inputer_config": {"use_cls_token": False, "use_sep_token": True}},
                        user_config={"num_attention_heads": meta["heads"], "attention_dropout": 0.0,
                                     "inputer_config": {"use_cls_token": False, "use_sep_token": False}})
        lc.set_component_classes(ops["Attention"], ops["Attention"], preds["Dot"])
    lc.set_item_ut(item_ut, ["title@glove", "category"])
    lc.set_user_ut(user_ut, ["history"])
    lc.set_column_map(ColumnMap(item_col="item_id", user_col="user_id", history_col="history", neg_col="neg",
                                label_col="click", group_col="user_id"))
    eh = EmbeddingHub(embedding_dim=D, transformation="auto", transformation_dropout=0.0)
    if meta["embed"] == "glove":
        eh.load_pretrained_embedding(None, vocab_name="glove", frozen=frozen,
                                     array=P["embedding_vocab_table.glove.embedding.weight"])
    eh.register_ut(item_ut, ["title@glove", "category"])
    lc.set_embedding_hub(eh)
    lc.build_components()
    lc.register_inputer_vocabs()
    model = Legommender(lc).to(dev)
    missing, unexpected = model.load_state_dict({k: torch.tensor(v) for k, v in P.items()}, strict=False)
    assert not unexpected and all(m.startswith("_") for m in missing), (missing, unexpected)
    assert set(P) == set(model.state_dict()), set(P) ^ set(model.state_dict())      # identical key set to the reference
    tb = ItemTables(tables["title_tok"], tables["title_len"], tables["cat"], dev)
    model.attach_item_table(tb)
    ids = {"item_id": torch.tensor(batch["cand"]), "history": torch.tensor(batch["hist"]),
           "__clicks_mask__": (torch.arange(50)[None] < torch.tensor(batch["hist_len"])[:, None]).long()}
    return model, ids, tb, G, logits, loss


def _check_grads(model, G, name):
    gscale = max(float(np.abs(g).max()) for g in G.values())
    got = dict(model.named_parameters())
    for k, g in G.items():
        d = got[k].grad.detach().cpu().numpy().astype(np.float64) - g
        assert float(np.abs(d).max()) <= 2e-3 * float(np.abs(g).max()) + 2e-7 * gscale, (name, k, float(np.abs(d).max()))
        assert float(np.linalg.norm(d)) <= 3e-4 * float(np.linalg.norm(g)) + 2e-7 * gscale * np.sqrt(d.size), (name, k)


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["naml_glove_d64", "nrms_null_d64", "nrms_glove_d64"])
@pytest.mark.parametrize("route", ["plugin", "engine"])
def test_legommender_routes_match_reference(name, route):
    from legommenders_amd.loader.env import Env
    dev = torch.device("cuda:0")
    model, ids, tb, G, logits, loss = _build(name, dev)
    if route == "engine":
        model.attach_engine(tb, B=ids["item_id"].shape[0])
    Env.train()
    model.train()
    out = model(batch=dict(ids))
    assert abs(float(out) - loss) < 2e-5
    out.backward()
    _check_grads(model, G, name)
    Env.test()
    model.eval()
    with torch.no_grad():
        scores = model(batch=dict(ids))
    assert float(np.abs(scores.cpu().numpy() - logits).max()) < 1e-4


@pytest.mark.gpu
@pytest.mark.parametrize("nested", [False, True])
def test_id_based_model_matches_reference(nested):
    """`use_item_content: false` (config/model/naml_id.yaml; model/legommender.py:237-248, lego_config.py:120-192): no item operator,
    candidates and clicked items are embeddings of their item ids, AdaOperator pools the history -- loss, every gradient (the
    [n_items, D] id table included) and the test-phase logits against the fixture generated from the reference
    (tests/golden/make_golden_id.py).  `nested`: the history as the reference's resampler ships it (the user inputer's nested sample,
    pads UNSET) instead of an id tensor."""
    from legommenders_amd.loader.class_hub import ClassHub
    from legommenders_amd.loader.column_map import ColumnMap
    from legommenders_amd.loader.embedding_hub import EmbeddingHub
    from legommenders_amd.loader.env import Env
    from legommenders_amd.loader.tables import Feature, Table, Vocab
    from legommenders_amd.model.lego_config import LegoConfig
    from legommenders_amd.model.legommender import Legommender
    dev = torch.device("cuda:0")
    meta, P, G, tables, batch, logits, loss = load_model_fixture("naml_id_d64")
    assert meta["kind"] == "naml_id"
    Env.set_device(dev)
    D = meta["D"]
    n_items = tables["title_tok"].shape[0]
    item_v, user_v = Vocab("item_id", n_items), Vocab("user_id", tables["user_hist"].shape[0])
    item_ut = Table([Feature("item_id", item_v), Feature("title@glove", Vocab("glove", meta["V"]), 30), Feature("category", Vocab("category", 18))],
                    {"item_id": np.arange(n_items), "title@glove": (tables["title_tok"], tables["title_len"]), "category": tables["cat"]}, "item_id")
    user_ut = Table([Feature("user_id", user_v), Feature("history", item_v, 50)],
                    {"user_id": np.arange(user_v.size), "history": (tables["user_hist"], tables["user_hist_len"])}, "user_id")
    ops, preds = ClassHub.operators(), ClassHub.predictors()
    lc = LegoConfig(hidden_size=D, item_hidden_size=D, neg_count=4, use_item_content=False,
                    user_config={"inputer_config": {"use_cls_token": False, "use_sep_token": False}})
    lc.set_component_classes(None, ops["Ada"], preds["Dot"])
    lc.set_item_ut(item_ut, ["title@glove", "category"])
    lc.set_user_ut(user_ut, ["history"])
    lc.set_column_map(ColumnMap(item_col="item_id", user_col="user_id", history_col="history", neg_col="neg", label_col="click", group_col="user_id"))
    eh = EmbeddingHub(embedding_dim=D, transformation="auto", transformation_dropout=0.0)
    eh.register_vocab(item_v)                                    # loader/manager.py:323-324
    lc.set_embedding_hub(eh)
    lc.build_components()
    lc.register_inputer_vocabs()
    assert lc.item_operator is None
    model = Legommender(lc).to(dev)
    assert set(P) == set(model.state_dict()), set(P) ^ set(model.state_dict())
    model.load_state_dict({k: torch.tensor(v) for k, v in P.items()})
    mask = (torch.arange(50)[None] < torch.tensor(batch["hist_len"])[:, None]).long()
    hist = torch.tensor(batch["hist"])
    if nested:
        hist = {"input_ids": {"history": torch.where(mask > 0, hist, torch.full_like(hist, -1))}, "attention_mask": mask}
    ids = {"item_id": torch.tensor(batch["cand"]), "history": hist, "__clicks_mask__": mask}
    Env.train()
    model.train()
    out = model(batch=dict(ids))
    assert abs(float(out) - loss) < 2e-5 * max(1.0, abs(loss))
    out.backward()
    _check_grads(model, G, "naml_id_d64")
    Env.test()
    model.eval()
    with torch.no_grad():
        scores = model(batch=dict(ids))
    assert float(np.abs(scores.cpu().numpy() - logits).max()) < 1e-4 * max(1.0, float(np.abs(logits).max()))


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["naml_glove_d64", "nrms_glove_d64"])
def test_unfrozen_pretrained_table_matches_oracle(name):
    """`load_pretrained_embedding(..., frozen=False)` (loader/embedding_hub.py:171,262: the GloVe table fine-tunes): the plug-in
    route's loss and EVERY gradient, the dense [V, 300] table gradient included, against the oracle with the table trainable"""
    from legommenders_amd.loader.env import Env
    from oracle import lego_oracle as O
    dev = torch.device("cuda:0")
    model, ids, tb, G, logits, loss = _build(name, dev, frozen=False)
    meta, P, _, tables, batch, _, _ = load_model_fixture(name)
    tk = "embedding_vocab_table.glove.embedding.weight"
    assert dict(model.named_parameters())[tk].requires_grad
    Env.train()
    model.train()
    out = model(batch=dict(ids))
    out.backward()
    ref_logits, ref_loss, ref_g = O.loss_and_grads(meta["kind"], P, {k: tables[k].astype("int64") for k in ("title_tok", "title_len", "cat")},
                                                   batch["cand"].astype("int64"), batch["hist"].astype("int64"),
                                                   batch["hist_len"].astype("int64"), heads=meta.get("heads", 8), glove=True, train_table=True)
    assert abs(float(out) - ref_loss) < 2e-5 and abs(ref_loss - loss) < 2e-5
    assert tk in ref_g and float(np.abs(ref_g[tk]).max()) > 0
    _check_grads(model, ref_g, name + " (table un-frozen)")
    # rows no token of the batch touches keep an exactly zero gradient (dense gradient, as the reference's autograd gives)
    gt = dict(model.named_parameters())[tk].grad.cpu().numpy()
    assert np.array_equal(np.abs(gt).sum(1) == 0, np.abs(ref_g[tk]).sum(1) == 0)


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["naml_glove_d64", "nrms_null_d64"])
def test_one_item_operator_call_equals_two(name):
    """id-only batch on the plug-in route: candidates and live history slots through ONE item-operator call
    (`Legommender._encode_items_once`) against the reference's control flow (one call for the candidates, one for the history):
    the same loss, gradients and scores"""
    from legommenders_amd.loader.env import Env
    dev = torch.device("cuda:0")
    res = {}
    for one in (True, False):
        model, ids, tb, G, logits, loss = _build(name, dev)
        model.one_item_call = one
        assert model._one_call_ok(dict(ids)) == one
        Env.train()
        model.train()
        out = model(batch=dict(ids))
        out.backward()
        _check_grads(model, G, f"{name} one_call={one}")
        res[one] = float(out)
        assert abs(res[one] - loss) < 2e-5
        Env.test()
        model.eval()
        with torch.no_grad():
            scores = model(batch=dict(ids))
        assert float(np.abs(scores.cpu().numpy() - logits).max()) < 1e-4
    assert abs(res[True] - res[False]) < 2e-6


def test_class_hub_discovers_reference_names():
    from legommenders_amd.loader.class_hub import ClassHub
    ops, preds = ClassHub.operators(), ClassHub.predictors()
    assert {"cnn", "ada", "attention"} <= set(ops.list()) and "dot" in preds
    assert ops["CNN"].__name__ == "CNNOperator" and preds("Dot").__name__ == "DotPredictor"


def test_cpu_device_is_refused():
    from legommenders_amd._lib import LegoHipError
    from legommenders_amd.loader.env import Env
    from legommenders_amd import functional as F_hip
    with pytest.raises(LegoHipError):
        Env.set_device(-1)
    with pytest.raises(LegoHipError):
        F_hip.linear(torch.zeros(2, 4), torch.zeros(3, 4))


def test_concat_sample_rebuilder_matches_reference_layout():
    """ConcatInputer.sample_rebuilder vs the oracle's vectorised restatement (pinned to the reference)."""
    from legommenders_amd.loader.tables import Feature, Table, Vocab
    from legommenders_amd.model.inputer.concat_inputer import ConcatInputer
    from oracle import lego_oracle as O
    tok = np.array([[5, 6, -1, -1], [7, -1, -1, -1], [1, 2, 3, 4]])
    tl = np.array([2, 1, 4])
    cat = np.array([9, 3, 0])
    ut = Table([Feature("item_id", Vocab("item_id", 3)), Feature("title", Vocab("glove", 10), 4),
                Feature("category", Vocab("category", 18))],
               {"item_id": np.arange(3), "title": (tok, tl), "category": cat}, "item_id")
    inp = ConcatInputer(use_cls_token=False, use_sep_token=True, ut=ut, inputs=["title", "category"], eh=None)
    t, c, s, m = O.concat_layout(torch.tensor(tok), torch.tensor(tl), torch.tensor(cat))
    for i in range(3):
        r = inp(ut[i])
        assert r["input_ids"]["title"].tolist() == t[i].tolist()
        assert r["input_ids"]["category"].tolist() == c[i].tolist()
        assert r["input_ids"][inp.vocab.name].tolist() == s[i].tolist()
        assert r["attention_mask"].tolist() == m[i].tolist()
