"""`torch.ops.lego_hip.*` (north star: kernels "registered as PyTorch-ROCm custom ops"):
schemas are registered without a GPU and have no CPU kernel; on the GPU every op passes `torch.library.opcheck`
(schema, autograd registration, fake tensor, AOT dispatch) and the differentiable ones agree with plain torch fp32."""
import pytest
import torch

import legommenders_amd.ops as lops

L = torch.ops.lego_hip


def test_ops_are_registered_with_the_dispatcher():
    for name in lops.OPS:
        op = getattr(L, name)
        assert op.default._schema.name == f"lego_hip::{name}"
    assert {"gather_rows", "linear", "conv3_relu_mask", "additive_pool", "mhsa", "dot_ce", "adam_step", "sample_negatives"} <= set(lops.OPS)
    s = str(L.adam_step.default._schema)
    assert "Tensor(a0!) p" in s and "Tensor(a3!) v" in s               # the in-place optimiser step declares its mutation
    assert "SymInt seed, SymInt site" in str(L.glove_project.default._schema)   # the Philox stream is an explicit argument


def test_no_cpu_kernel():
    with pytest.raises(NotImplementedError):
        L.linear(torch.zeros(2, 4), torch.zeros(3, 4), None)
    from legommenders_amd import functional as F
    from legommenders_amd._lib import LegoHipError
    with pytest.raises(LegoHipError):
        F.additive_attention(torch.zeros(2, 3, 4), torch.ones(2, 3), torch.zeros(4, 4), torch.zeros(4), torch.zeros(1, 4))


def test_fake_implementations_trace_without_a_gpu():
    from torch._subclasses.fake_tensor import FakeTensorMode
    with FakeTensorMode():
        x = torch.empty(6, 5, 16, device="cuda")
        m = torch.ones(6, 5, dtype=torch.int64, device="cuda")
        W1, b1, w2 = torch.empty(8, 16, device="cuda"), torch.empty(8, device="cuda"), torch.empty(1, 8, device="cuda")
        out, t, w = L.additive_pool(x, m, W1, b1, w2)
        assert out.shape == (6, 16) and t.shape == (30, 8) and w.shape == (30,)
        y = L.conv3_relu_mask(x, m, torch.empty(32, 16, 3, device="cuda"), torch.empty(32, device="cuda"), 0.1, 1, 2)
        assert y.shape == (6, 5, 32)
        loss, scores = L.dot_ce(torch.empty(6, 16, device="cuda"), torch.empty(6, 5, 16, device="cuda"))
        assert loss.shape == () and scores.shape == (6, 5)


# ----------------------------------------------------------------------------------------------- GPU
def _dev():
    assert torch.cuda.is_available()
    return torch.device("cuda:0")


def _r(*shape, seed=0, grad=False, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    t = (torch.randn(*shape, generator=g) * scale).to(_dev())
    return t.requires_grad_(grad)


def _mask(n, L_, seed=0):
    g = torch.Generator().manual_seed(seed)
    lens = torch.randint(1, L_ + 1, (n,), generator=g)
    return (torch.arange(L_)[None] < lens[:, None]).long().to(_dev())


def _samples():
    dev = _dev()
    n, Ls, D, A = 6, 7, 32, 16
    ids = torch.tensor([3, -1, 0, 7, 7, 2, -1, 5], device=dev)
    return {
        "gather_rows": (_r(9, 12, grad=True), ids),
        "scatter_add_rows": (_r(8, 12), ids, 9),
        "linear": (_r(10, 12, grad=True), _r(8, 12, seed=1, grad=True), _r(8, seed=2, grad=True)),
        "linear_bwd": (_r(10, 8), _r(10, 12, seed=1), _r(8, 12, seed=2), True),
        "glove_project": (ids, _r(9, 12), _r(32, 12, seed=1, grad=True), _r(32, seed=2, grad=True), 0.1, 7, 3),
        "glove_project_bwd": (_r(8, 32), _r(8, 12, seed=1), ids, _r(32, 12, seed=2), 0.1, 7, 3),
        "glove_project_bwd_table": (_r(8, 32), ids, _r(32, 12, seed=2), 9, 0.1, 7, 3),
        "conv3_relu_mask": (_r(n, Ls, D, grad=True), _mask(n, Ls), _r(D, D, 3, seed=1, grad=True, scale=0.1), _r(D, seed=2, grad=True), 0.0, 0, 0),
        "additive_pool": (_r(n, Ls, D, grad=True), _mask(n, Ls), _r(A, D, seed=1, grad=True, scale=0.2), _r(A, seed=2, grad=True),
                          _r(1, A, seed=3, grad=True)),
        "mhsa": (_r(n, Ls, D, grad=True), _mask(n, Ls), _r(3 * D, D, seed=1, grad=True, scale=0.2), _r(3 * D, seed=2, grad=True),
                 _r(D, D, seed=3, grad=True, scale=0.2), _r(D, seed=4, grad=True), 4, 0.0, 0, 0),
        "rowdot": (_r(10, D, grad=True), _r(10, D, seed=1, grad=True)),
        "rowdot_bwd": (_r(10), _r(10, D, seed=1), _r(10, D, seed=2)),
        "dot_ce": (_r(n, D, grad=True), _r(n, 5, D, seed=1, grad=True)),
        "sample_negatives": (torch.tensor([0, 1, 2, 1], dtype=torch.int32, device=dev), torch.tensor([5, 6, 7, 8], dtype=torch.int32, device=dev),
                             torch.arange(30, dtype=torch.int32, device=dev).view(3, 10), torch.tensor([10, 2, 0], dtype=torch.int32, device=dev),
                             4, 50, 11, 0, 0, 1),
        "gather_history": (torch.tensor([2, 0], dtype=torch.int32, device=dev), torch.arange(15, dtype=torch.int32, device=dev).view(3, 5),
                           torch.tensor([5, 1, 3], dtype=torch.int32, device=dev)),
    }


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["gather_rows", "scatter_add_rows", "linear", "linear_bwd", "glove_project", "glove_project_bwd", "glove_project_bwd_table",
                                  "conv3_relu_mask", "additive_pool", "mhsa", "rowdot", "rowdot_bwd", "dot_ce", "sample_negatives",
                                  "gather_history"])
def test_opcheck(name):
    args = _samples()[name]
    # gradcheck-style numerics are covered below against torch; opcheck holds the registrations themselves
    torch.library.opcheck(getattr(L, name).default, args,
                          test_utils=("test_schema", "test_autograd_registration", "test_faketensor", "test_aot_dispatch_dynamic"))


@pytest.mark.gpu
def test_adam_step_op_mutates_in_place_and_matches_torch():
    p0 = _r(1000)
    g0 = _r(1000, seed=1)
    ref = p0.clone().requires_grad_(True)
    opt = torch.optim.Adam([ref], lr=1e-2)
    p, m, v = p0.clone(), torch.zeros_like(p0), torch.zeros_like(p0)
    for step in range(1, 4):
        ref.grad = g0.clone() * step
        opt.step()
        g = g0.clone() * step
        L.adam_step(p, g, m, v, 1e-2, step, 1.0, True)
        assert float(g.abs().max()) == 0.0
    assert float((p - ref.detach()).abs().max()) < 1e-6
    torch.library.opcheck(L.adam_step.default, (p.clone(), g0.clone(), m.clone(), v.clone(), 1e-2, 4, 1.0, False),
                          test_utils=("test_schema", "test_faketensor"))


@pytest.mark.gpu
def test_ops_autograd_matches_torch_fp32():
    """the dispatcher route end to end: a NAML-shaped stack built from torch.ops.lego_hip.* vs the same stack in plain torch"""
    import torch.nn.functional as TF
    n, Ls, D, A = 6, 7, 32, 16
    h, mask = _r(n, Ls, D, grad=True), _mask(n, Ls)
    w, b = _r(D, D, 3, seed=1, grad=True, scale=0.1), _r(D, seed=2, grad=True)
    W1, b1, w2 = _r(A, D, seed=3, grad=True, scale=0.2), _r(A, seed=4, grad=True), _r(1, A, seed=5, grad=True)
    Wl, bl = _r(D, D, seed=6, grad=True, scale=0.2), _r(D, seed=7, grad=True)
    leaves = [h, w, b, W1, b1, w2, Wl, bl]

    def ours():
        y = L.conv3_relu_mask(h, mask, w, b, 0.0, 0, 0)
        pooled = L.additive_pool(y, mask, W1, b1, w2)[0]
        z = L.linear(pooled, Wl, bl)
        user, items = z[:2], z[2:].reshape(2, 2, D)
        loss, scores = L.dot_ce(user, items)
        return loss, scores

    def ref():
        m = mask.float()
        y = TF.relu(TF.conv1d(h.transpose(1, 2), w, b, padding="same").transpose(1, 2)) * m[..., None]
        a = torch.tanh(y @ W1.t() + b1) @ w2.t()
        e = torch.exp(a.squeeze(-1)) * m
        wgt = e / (e.sum(1, keepdim=True) + 2.0 ** -23)
        pooled = (wgt[..., None] * y).sum(1)
        z = pooled @ Wl.t() + bl
        user, items = z[:2], z[2:].reshape(2, 2, D)
        scores = (user[:, None] * items).sum(-1)
        return TF.cross_entropy(scores, torch.zeros(2, dtype=torch.long, device=scores.device)), scores

    lo, so = ours()
    go = torch.autograd.grad(lo, leaves)
    lr_, sr = ref()
    gr = torch.autograd.grad(lr_, leaves)
    assert float((so - sr).abs().max()) < 1e-4 and abs(float(lo) - float(lr_)) < 1e-5
    for a, b_ in zip(go, gr):
        assert float((a - b_).abs().max()) <= 2e-4 * float(b_.abs().max()) + 1e-6


@pytest.mark.gpu
def test_mhsa_op_with_every_position_masked():
    """a batch whose sequences are all empty (R == 0 live rows) through the dispatcher route: zeros out, zero gradients, no error
    (lse[:0] is a NULL pointer and the entry point refuses NULL lse + NULL probs)"""
    n, Ls, D, heads = 3, 5, 32, 4
    x = _r(n, Ls, D, grad=True)
    mask = torch.zeros(n, Ls, dtype=torch.int32, device=x.device)
    in_w, in_b = _r(3 * D, D, seed=1, grad=True, scale=0.2), _r(3 * D, seed=2, grad=True)
    out_w, out_b = _r(D, D, seed=3, grad=True, scale=0.2), _r(D, seed=4, grad=True)
    y = L.mhsa(x, mask, in_w, in_b, out_w, out_b, heads, 0.1, 11, 3)[0]
    assert y.shape == (n, Ls, D) and float(y.abs().max()) == 0.0
    g = torch.autograd.grad(y.sum(), [x, in_w, in_b, out_w, out_b])
    assert all(float(t.abs().max()) == 0.0 for t in g)


@pytest.mark.gpu
def test_plugin_operators_dispatch_through_torch_ops():
    """model.operators reach the kernels through the dispatcher: a dispatch-mode trace of CNNOperator's forward sees lego_hip ops"""
    from torch.utils._python_dispatch import TorchDispatchMode
    from legommenders_amd import functional as F
    seen = []

    class Rec(TorchDispatchMode):
        def __torch_dispatch__(self, func, types, args=(), kwargs=None):
            seen.append(str(func))
            return func(*args, **(kwargs or {}))
    n, Ls, D = 4, 5, 32
    with Rec():
        y = F.conv3_relu_mask(_r(n, Ls, D), _mask(n, Ls), _r(D, D, 3, seed=1), _r(D, seed=2))
        F.additive_attention(y, _mask(n, Ls), _r(16, D, seed=3), _r(16, seed=4), _r(1, 16, seed=5))
    assert any("lego_hip.conv3_relu_mask" in s for s in seen) and any("lego_hip.additive_pool" in s for s in seen)
