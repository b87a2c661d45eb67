"""The oracle (oracle/lego_oracle.py) against the golden vectors generated from the real reference."""
import os

import numpy as np
import pytest
import torch

from oracle import lego_oracle as O
from tests.golden_util import GOLDEN, MODEL_FIXTURES, load_model_fixture


def _ops():
    return np.load(os.path.join(GOLDEN, "ops.npz"))


def test_additive_attention_fwd_bwd():
    z = _ops()
    x = torch.tensor(z["add.x"], requires_grad=True)
    W1 = torch.tensor(z["add.W1"], requires_grad=True)
    b1 = torch.tensor(z["add.b1"], requires_grad=True)
    w2 = torch.tensor(z["add.w2"], requires_grad=True)
    y = O.additive_attention(x, torch.tensor(z["add.mask"]), W1, b1, w2)
    np.testing.assert_allclose(y.detach().numpy(), z["add.y"], rtol=1e-5, atol=1e-6)
    assert np.all(y.detach().numpy()[0] == 0.0)          # all-masked row -> exact zeros
    y.backward(torch.tensor(z["add.gy"]))
    np.testing.assert_allclose(x.grad.numpy(), z["add.gx"], rtol=1e-4, atol=1e-6)
    np.testing.assert_allclose(W1.grad.numpy(), z["add.gW1"], rtol=1e-4, atol=1e-6)
    np.testing.assert_allclose(b1.grad.numpy(), z["add.gb1"], rtol=1e-4, atol=1e-6)
    np.testing.assert_allclose(w2.grad.numpy(), z["add.gw2"], rtol=1e-4, atol=1e-6)


def test_mhsa_fwd_bwd():
    z = _ops()
    x = torch.tensor(z["mha.x"], requires_grad=True)
    ps = [torch.tensor(z[k], requires_grad=True) for k in ("mha.in_w", "mha.in_b", "mha.out_w", "mha.out_b")]
    mask = torch.tensor(z["mha.mask"])
    y = O.mhsa(x, mask, *ps, int(z["mha.heads"]))
    live = z["mha.mask"].astype(bool)
    np.testing.assert_allclose(y.detach().numpy()[live], z["mha.y"][live], rtol=1e-4, atol=2e-6)
    y.backward(torch.tensor(z["mha.gy"]))
    np.testing.assert_allclose(x.grad.numpy(), z["mha.gx"], rtol=1e-4, atol=2e-6)
    for p, k in zip(ps, ("mha.gin_w", "mha.gin_b", "mha.gout_w", "mha.gout_b")):
        np.testing.assert_allclose(p.grad.numpy(), z[k], rtol=1e-4, atol=2e-6)


def test_dot_ce():
    z = _ops()
    u = torch.tensor(z["dot.u"], requires_grad=True)
    it = torch.tensor(z["dot.i"], requires_grad=True)
    s = O.dot_scores(u, it)
    np.testing.assert_allclose(s.detach().numpy(), z["dot.s"], rtol=1e-5, atol=1e-5)
    loss = O.ce_label0(s)
    assert abs(float(loss) - float(z["dot.loss"])) < 1e-6
    loss.backward()
    np.testing.assert_allclose(u.grad.numpy(), z["dot.gu"], rtol=1e-4, atol=1e-6)
    np.testing.assert_allclose(it.grad.numpy(), z["dot.gi"], rtol=1e-4, atol=1e-6)


def test_adam_linear_schedule_trajectory():
    z = _ops()
    p = z["adam.traj"][0]
    m = np.zeros_like(p)
    v = np.zeros_like(p)
    total = int(z["adam.total"])
    for step in range(3):
        lr = float(z["adam.lr"]) * O.linear_schedule_factor(step, total)
        p, m, v = O.adam_step(p, z["adam.g"][step], m, v, step + 1, lr)
        np.testing.assert_allclose(p, z["adam.traj"][step + 1], rtol=1e-6, atol=1e-7)


@pytest.mark.parametrize("fixture", ["metrics.npz", "metrics_full.npz"])
def test_metrics_match_metricpool(fixture):
    z = np.load(os.path.join(GOLDEN, fixture))
    got = O.grouped_metrics(z["scores"], z["labels"], z["groups"], names=[str(n) for n in z["names"]])
    for n, v in zip(z["names"], z["values"]):
        assert abs(got[str(n)] - float(v)) < 5e-7, (n, got[str(n)], v)


@pytest.mark.parametrize("name", MODEL_FIXTURES)
def test_full_forward_backward(name):
    meta, P, G, tables, batch, logits, loss = load_model_fixture(name)
    got_logits, got_loss, grads = O.loss_and_grads(
        meta["kind"], P, tables, batch["cand"], batch["hist"], batch["hist_len"],
        heads=meta["heads"], glove=(meta["embed"] == "glove"))
    np.testing.assert_allclose(got_logits, logits, rtol=1e-5, atol=2e-6)
    assert abs(got_loss - loss) < 2e-6
    assert set(G) <= set(grads), set(G) - set(grads)
    for k, g in G.items():
        scale = max(1e-8, float(np.abs(g).max()))
        assert float(np.abs(grads[k] - g).max()) <= 2e-4 * scale + 1e-9, k


def test_concat_layout_edges():
    tok = torch.tensor([[5, 6, -1, -1], [7, -1, -1, -1], [1, 2, 3, 4]])
    tl = torch.tensor([2, 1, 4])
    cat = torch.tensor([9, 3, 0])
    t, c, s, m = O.concat_layout(tok, tl, cat, use_sep=True)
    assert t.shape == (3, 7)
    assert t[0].tolist() == [5, 6, -1, -1, -1, -1, -1]
    assert c[0].tolist() == [-1, -1, -1, 9, -1, -1, -1]
    assert s[0].tolist() == [-1, -1, 2, -1, 2, 0, 0]
    assert m[0].tolist() == [1, 1, 1, 1, 1, 0, 0]
    assert m[2].tolist() == [1] * 7 and s[2].tolist() == [-1, -1, -1, -1, 2, -1, 2]


@pytest.mark.parametrize("fixture", ["metrics.npz", "metrics_full.npz"])
def test_product_metrics_match_metricpool_and_oracle(fixture):
    """legommenders_amd.metrics (host form) against the reference MetricPool fixtures (every metric of
    MetricPool.metric_list in metrics_full.npz) and the oracle."""
    from legommenders_amd import metrics as PM
    z = np.load(os.path.join(GOLDEN, fixture))
    names = [str(n) for n in z["names"]]
    got = PM.calculate(z["scores"], z["labels"], z["groups"], names)
    ref = O.grouped_metrics(z["scores"], z["labels"], z["groups"], names=names)
    for n, v in zip(names, z["values"]):
        assert abs(got[n] - float(v)) < 5e-7 and abs(got[n] - ref[n]) < 1e-7, n


def test_bert_news_encoder_full_forward_backward():
    """the oracle's restatement of the BERT news encoder path (transformers' BertModel arithmetic +
    once_operator.py's Linear + additive pool, with the tune_from = 0 layer slicing) against the reference run."""
    meta, P, G, tables, batch, logits, loss = load_model_fixture("bert_naml_small")
    lg, ls, g = O.loss_and_grads("bert_naml", P, tables, batch["cand"], batch["hist"], batch["hist_len"], heads=meta["heads"],
                                 frozen=("embedding_vocab_table.glove.weight",), bert_layers=meta["layers_kept"],
                                 bert_eps=meta["bert"]["layer_norm_eps"])
    assert meta["layers_kept"] == meta["bert"]["num_hidden_layers"] - 1          # the reference dropped block 0
    np.testing.assert_allclose(lg, logits, rtol=1e-5, atol=2e-6)
    assert abs(ls - loss) < 2e-6
    gscale = max(float(np.abs(v).max()) for v in G.values())
    assert len(G) == 45
    for k, v in G.items():
        assert float(np.abs(g[k] - v).max()) <= 1e-4 * float(np.abs(v).max()) + 1e-6 * gscale, k


def test_bert_cached_layer_mode_full_forward_backward():
    """`tune_from = 1` (cached-layer mode, once_operator.py:99-134,182-188): the oracle rebuilds the reference's layer-1
    cache from the checkpoint the reference loaded, then reproduces its logits / loss / gradients from that cache."""
    import json
    z = np.load(os.path.join(GOLDEN, "bert_naml_tune1.npz"))
    meta, P, G, tables, batch, logits, loss = load_model_fixture("bert_naml_tune1")
    assert meta["tune_from"] == 1 and meta["layers_kept"] == meta["bert"]["num_hidden_layers"] - 2    # block 1 is skipped too
    ckpt = {k[len("ckpt::"):]: torch.tensor(z[k]) for k in z.files if k.startswith("ckpt::")}
    hidden, mask = O.bert_layer_cache(ckpt, torch.tensor(P["embedding_vocab_table.glove.weight"]),
                                      torch.tensor(P["embedding_vocab_table.category.weight"]), torch.tensor(tables["title_tok"]),
                                      torch.tensor(tables["title_len"]), torch.tensor(tables["cat"]), layer=1,
                                      n_layers=meta["bert"]["num_hidden_layers"], heads=meta["heads"],
                                      eps=meta["bert"]["layer_norm_eps"])
    assert np.array_equal(mask.numpy(), z["cache::mask"])
    np.testing.assert_allclose(hidden.numpy(), z["cache::hidden"], rtol=1e-5, atol=2e-6)
    lg, ls, g = O.loss_and_grads("bert_naml", P, tables, batch["cand"], batch["hist"], batch["hist_len"], heads=meta["heads"],
                                 frozen=("embedding_vocab_table.glove.weight", "embedding_vocab_table.category.weight"),
                                 bert_layers=meta["layers_kept"], bert_eps=meta["bert"]["layer_norm_eps"],
                                 layer_cache=(z["cache::hidden"], z["cache::mask"]))
    np.testing.assert_allclose(lg, logits, rtol=1e-5, atol=2e-6)
    assert abs(ls - loss) < 2e-6
    assert set(G) <= set(g)
    for k, v in G.items():
        assert float(np.abs(g[k] - v).max()) <= 2e-4 * max(1e-8, float(np.abs(v).max())) + 1e-9, k
