"""The OPT-IN split-bf16 product mode (include/lego_hip.h: lego_set_product_mode).  It is not the parity mode --
every other test of this suite runs in exact f32 -- so what is held here is what the mode promises: each large product within a few
1e-6 of the float64 result (relative to the largest output), the models' logits within the north star's 1e-3 of the oracle (measured:
~1e-5), gradients within 5e-3 in Frobenius norm, and (tests/test_train_band.py) the trained GAUC inside the reference's band."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture
def split_mode():
    from legommenders_amd import _lib
    assert _lib.product_mode() == _lib.EXACT_F32          # the suite's default
    _lib.set_product_mode(_lib.SPLIT_BF16)
    try:
        yield
    finally:
        _lib.set_product_mode(_lib.EXACT_F32)


def _rel(got, ref):
    return float((got.double().cpu() - ref).abs().max() / ref.abs().max())


def _products(dev, seed):
    from legommenders_amd import kernels as K
    g = torch.Generator().manual_seed(seed)
    out = {}
    for (M, N, Kd) in [(4100, 256, 768), (2500, 256, 300), (3000, 768, 256), (2200, 64, 128)]:
        x, W, b = torch.randn(M, Kd, generator=g), torch.randn(N, Kd, generator=g) * 0.1, torch.randn(N, generator=g)
        gy = torch.randn(M, N, generator=g)
        xd, Wd, bd, gd = x.to(dev), W.to(dev), b.to(dev), gy.to(dev)
        dW = torch.zeros(N, Kd, device=dev)
        K.linear_bwd_weight(gd, xd, dW)
        out[(M, N, Kd)] = ((K.linear_fwd(xd, Wd, bd, act=0), x.double() @ W.double().T + b.double()),
                           (K.linear_bwd_data(gd, Wd), gy.double() @ W.double()),
                           (dW, gy.double().T @ x.double()))
    # the direct three-tap conv (the Winograd entry points are exact-mode only)
    n, L, D = 120, 30, 256
    lens = torch.randint(1, L + 1, (n,), generator=g)
    mask = (torch.arange(L)[None] < lens[:, None]).int()
    h = torch.randn(n, L, D, generator=g) * mask[..., None]
    w, b = torch.randn(D, D, 3, generator=g) * 0.05, torch.randn(D, generator=g) * 0.1
    ref = torch.relu(torch.nn.functional.conv1d(h.double().permute(0, 2, 1), w.double(), b.double(), padding="same").permute(0, 2, 1)) * mask[..., None]
    plan = K.plan_dense(mask.to(dev))
    y = K.conv3_fwd(h.reshape(n * L, D).to(dev), K.conv3_pack(w.to(dev)), b.to(dev), plan)
    out["conv3_fwd"] = ((y.view(n, L, D), ref),)
    return out


def test_split_products_against_float64(split_mode):
    from legommenders_amd import _lib
    dev = torch.device("cuda:0")
    split = _products(dev, 3)
    _lib.set_product_mode(_lib.EXACT_F32)
    exact = _products(dev, 3)
    _lib.set_product_mode(_lib.SPLIT_BF16)
    for key in split:
        for (got_s, ref), (got_e, _) in zip(split[key], exact[key]):
            es, ee = _rel(got_s, ref), _rel(got_e, ref)
            assert ee < 3e-6, (key, "exact", ee)
            assert es < 2e-5, (key, "split", es)
            assert not torch.equal(got_s, got_e), (key, "the split route was not taken")


@pytest.mark.parametrize("kind", ["naml", "nrms"])
def test_split_mode_logits_and_gradients_at_headline_size(kind, split_mode):
    """BASELINE.json configs 2 / 3 at their batch shape, split mode against the float64-free oracle: logits far inside the north
    star's 1e-3, loss 1e-4, every gradient 5e-3 in Frobenius norm"""
    from oracle import lego_oracle as O
    from legommenders_amd import engine as E
    from legommenders_amd.synthetic import glove_like, init_naml_params, init_nrms_params, make_world
    dev = torch.device("cuda:0")
    D, B, C, S, V = 256, 64, 5, 50, 50000
    w = make_world(seed=2023, n_users=4000, n_rows=4000, V=V)
    if kind == "naml":
        P = init_naml_params(D=D, V=V, n_cat=w["n_cat"], glove=glove_like(V, 300, seed=2024, device=dev))
        okw = {}
    else:
        P = init_nrms_params(D=D, V=V, n_cat=w["n_cat"], heads=8, glove=None)
        okw = dict(heads=8, glove=False)
    Pd = {k: v.to(dev).contiguous() for k, v in P.items()}
    tb = E.ItemTables(w["title_tok"], w["title_len"], w["cat"], dev)
    eng = (E.NamlEngine(Pd, tb, B, C, S, p_proj=0.0, p_conv=0.0) if kind == "naml"
           else E.NrmsEngine(Pd, tb, B, C, S, heads=8, glove=False, p_proj=0.0, p_att=0.0))
    if kind == "naml":
        assert not eng.wino                                   # the direct conv: the Winograd kernels have no split form
    rs = np.random.RandomState(11)
    cand = rs.randint(0, w["n_items"], size=(B, C))
    users = rs.randint(0, 4000, size=B)
    hist, hl = w["user_hist"][users].copy(), np.maximum(w["user_hist_len"][users], 1)
    hist = hist * (np.arange(S)[None] < hl[:, None])
    ids = [torch.tensor(np.ascontiguousarray(a)).int().to(dev).contiguous() for a in (cand, hist, hl)]
    scores, loss = eng.forward(*ids, training=False)
    G = eng.grads_like()
    eng.backward(G)
    torch.cuda.synchronize()
    tables = {k: w[k].astype(np.int64) for k in ("title_tok", "title_len", "cat")}
    lg, ls, g = O.loss_and_grads(kind, {k: v.cpu().numpy() for k, v in P.items()}, tables, cand, hist, hl, **okw)
    err = float(np.abs(scores.cpu().numpy() - lg).max())
    print(kind, "split-bf16 max |logit - oracle| =", err, "loss diff", abs(float(loss) - ls))
    assert err < 1e-3 and abs(float(loss) - ls) < 1e-4
    gscale = max(float(np.abs(v).max()) for v in g.values())
    worst = ("", 0.0)
    for k, ref in g.items():
        d = G[k].cpu().numpy().astype(np.float64) - ref
        rel = float(np.linalg.norm(d) / max(np.linalg.norm(ref), 1e-30))
        worst = max(worst, (k, rel), key=lambda t: t[1]) if np.linalg.norm(ref) > 1e-6 * gscale * np.sqrt(d.size) else worst
        # 5e-3: a conv pre-activation within the mode's error of 0 flips its ReLU (a rank-1 change of the tensors upstream of it)
        assert np.linalg.norm(d) <= 5e-3 * np.linalg.norm(ref) + 1e-6 * gscale * np.sqrt(d.size), (k, np.linalg.norm(d), np.linalg.norm(ref))
    print(kind, "worst relative Frobenius gradient error:", worst)


def test_split_mode_bert_blocks_against_float64(split_mode):
    """the BERT blocks on the path's kernels (legommenders_amd/bert_native.py) in split mode, against the `transformers` module tree in
    float64: hidden states within 1e-4 of the largest, gradients within 2e-3 (2 450 live rows: every product takes the split route)"""
    import transformers
    from legommenders_amd import bert_native
    dev = torch.device("cuda:0")
    torch.manual_seed(9)
    cfg = transformers.BertConfig(vocab_size=50, hidden_size=128, num_hidden_layers=2, num_attention_heads=4, intermediate_size=512,
                                  max_position_embeddings=32, hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0)
    ref = transformers.BertModel(cfg, add_pooling_layer=False).double().eval()
    mine = transformers.BertModel(cfg, add_pooling_layer=False)
    mine.load_state_dict({k: v.float() for k, v in ref.state_dict().items()})
    mine = mine.to(dev).eval()
    n, L, H = 128, 24, 128
    lens = torch.randint(12, L + 1, (n,))
    mask = (torch.arange(L)[None] < lens[:, None]).long()
    assert int(mask.sum()) >= 2048
    x = torch.randn(n, L, H, dtype=torch.float64)
    w = torch.randn(n, L, H, dtype=torch.float64) * mask[..., None]
    xr = x.clone().requires_grad_(True)
    hr = ref(inputs_embeds=xr, attention_mask=mask).last_hidden_state
    (hr * w).sum().backward()
    xm = x.float().to(dev).requires_grad_(True)
    hm = bert_native.encoder_forward(mine, xm, mask.to(dev), True)
    (hm * w.float().to(dev)).sum().backward()
    live = mask.bool()
    from legommenders_amd import _lib
    _lib.set_product_mode(_lib.EXACT_F32)
    with torch.no_grad():
        h_exact = bert_native.encoder_forward(mine, xm.detach(), mask.to(dev), True)
    _lib.set_product_mode(_lib.SPLIT_BF16)
    assert not torch.equal(h_exact, hm.detach())                       # the split route was taken
    e = float((hm.detach().cpu().double()[live] - hr.detach()[live]).abs().max() / hr.detach()[live].abs().max())
    print("split-bf16 BERT blocks: relative hidden-state error", e)
    assert e < 1e-4
    pm = dict(mine.named_parameters())
    worst = 0.0
    gmax = max(float(p.grad.norm()) for p in ref.parameters() if p.grad is not None)
    for k, p in ref.named_parameters():
        if p.grad is None or float(p.grad.norm()) < 1e-9 * gmax:          # (a key bias cannot move a softmax: its gradient is rounding noise)
            continue
        d = float((pm[k].grad.cpu().double() - p.grad).norm() / p.grad.norm())
        worst = max(worst, d)
        assert d < 2e-3, (k, d)
    d = float((xm.grad.cpu().double() - xr.grad).norm() / xr.grad.norm())
    print("split-bf16 BERT blocks: worst relative gradient error", max(worst, d))
    assert d < 2e-3
