"""`torch.ops.lego_hip.*`: the HIP kernels of liblego_hip.so registered with the PyTorch dispatcher as custom ops
(SURVEY.md section 8b "Native/FFI layer"; north star: "hand-written CDNA4 HIP kernels registered as PyTorch-ROCm
custom ops").

Every forward op has a CUDA (= ROCm) implementation only -- a CPU tensor finds no kernel and raises, there is no
fallback -- a fake (meta) implementation so that `torch.compile` / `make_fx` / `torch.library.opcheck` can trace it, and
an autograd formula whose backward is itself built from registered ops (`*_bwd`), so the backward traces too.
Randomness is explicit: a dropout op takes `(p, seed, site)` and draws its keep bits from that Philox stream, forward
and backward alike; nothing reads hidden state.  The plug-in classes (`model/operators/*`, `model/predictors/*`,
`loader/embedding_hub`) reach the kernels through `legommenders_amd.functional`, which calls these ops; the fused
ragged engines (`engine.py`) stay the training fast path and call the same C ABI directly.

Reference call sites each op replaces are cited on the op.
"""
from __future__ import annotations

import ctypes
from typing import Optional, Tuple

import torch
from torch.library import custom_op, register_autograd, register_fake

from . import kernels as K
from ._lib import call

NS = "lego_hip"


def _drop(p: float, seed: int, site: int):
    return (float(p), int(seed), int(site)) if p > 0.0 else None


def _f(t: torch.Tensor) -> torch.Tensor:
    return t.detach().to(torch.float32).contiguous()


def _cuda(name, **kw):
    return custom_op(f"{NS}::{name}", mutates_args=kw.pop("mutates_args", ()), device_types="cuda", **kw)


# =============================================================================== embedding look-ups (a4)
@_cuda("gather_rows")
def gather_rows(table: torch.Tensor, idx: torch.Tensor) -> torch.Tensor:
    """out[r,:] = table[idx[r],:], idx < 0 -> zero row (nn.Embedding look-up behind the inputers' pad handling,
    loader/embedding_hub.py:73-96; the coalesced HBM row gather of the hot path)"""
    return K.gather_rows(_f(table), idx.to(torch.int32).contiguous())


@register_fake(f"{NS}::gather_rows")
def _(table, idx):
    return table.new_empty(idx.numel(), table.shape[1], dtype=torch.float32)


@_cuda("scatter_add_rows")
def scatter_add_rows(g: torch.Tensor, idx: torch.Tensor, rows: int) -> torch.Tensor:
    """dense [rows, W] gradient of gather_rows (embedding_hub.py:325-335: nn.Embedding's dense gradient)"""
    out = torch.zeros(rows, g.shape[1], dtype=torch.float32, device=g.device)
    return K.scatter_add_rows(out, idx.to(torch.int32).contiguous(), _f(g))


@register_fake(f"{NS}::scatter_add_rows")
def _(g, idx, rows):
    return g.new_empty(rows, g.shape[1], dtype=torch.float32)


def _gather_setup(ctx, inputs, output):
    table, idx = inputs
    ctx.save_for_backward(idx)
    ctx.rows = table.shape[0]


def _gather_bwd(ctx, g):
    (idx,) = ctx.saved_tensors
    return torch.ops.lego_hip.scatter_add_rows(g.contiguous(), idx, ctx.rows), None


register_autograd(f"{NS}::gather_rows", _gather_bwd, setup_context=_gather_setup)


# =============================================================================== nn.Linear (a4 / a5 / a8)
@_cuda("linear")
def linear(x: torch.Tensor, W: torch.Tensor, b: Optional[torch.Tensor]) -> torch.Tensor:
    """y = x W^T + b on the fp32 MFMA kernels (x: [M,K], W: [N,K]; nn.Linear of cnn_operator.py:58-60,
    attention_operator.py:52, embedding_hub.py:95)"""
    return K.linear_fwd(_f(x), _f(W), None if b is None else _f(b), act=0)


@register_fake(f"{NS}::linear")
def _(x, W, b):
    return x.new_empty(x.shape[0], W.shape[0], dtype=torch.float32)


@_cuda("linear_bwd")
def linear_bwd(g: torch.Tensor, x: torch.Tensor, W: torch.Tensor, has_bias: bool) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor]:
    g, x, W = _f(g), _f(x), _f(W)
    gW = torch.zeros_like(W)
    K.linear_bwd_weight(g, x, gW)
    gb = torch.zeros(W.shape[0], dtype=torch.float32, device=W.device)
    if has_bias:
        K.colsum(g, gb)
    return K.linear_bwd_data(g, W), gW, gb


@register_fake(f"{NS}::linear_bwd")
def _(g, x, W, has_bias):
    return torch.empty_like(x, dtype=torch.float32), torch.empty_like(W, dtype=torch.float32), W.new_empty(W.shape[0], dtype=torch.float32)


def _linear_setup(ctx, inputs, output):
    x, W, b = inputs
    ctx.save_for_backward(x, W)
    ctx.has_b = b is not None


def _linear_bwd(ctx, g):
    x, W = ctx.saved_tensors
    gx, gW, gb = torch.ops.lego_hip.linear_bwd(g.contiguous(), x, W, ctx.has_b)
    return gx, gW, (gb if ctx.has_b else None)


register_autograd(f"{NS}::linear", _linear_bwd, setup_context=_linear_setup)


# =============================================================================== frozen table -> projection (a3 / a4)
@_cuda("glove_project")
def glove_project(ids: torch.Tensor, table: torch.Tensor, W: torch.Tensor, b: torch.Tensor, p: float, seed: int,
                  site: int) -> Tuple[torch.Tensor, torch.Tensor]:
    """Transformation.forward + SimpleInputer masking: Dropout(Linear(Embedding(ids))) * (ids >= 0)
    (loader/embedding_hub.py:95-96, model/inputer/simple_inputer.py:55-63); ids flat [R], table frozen.
    Returns (H [R,D], X [R,E0] the gathered rows, kept for the weight gradient)."""
    flat = ids.to(torch.int32).contiguous()
    X = K.gather_rows(_f(table), flat)
    rowinfo = torch.where(flat >= 0, torch.full_like(flat, 4), torch.zeros_like(flat))
    H = K.linear_fwd(X, _f(W), _f(b), act=0, rowinfo=rowinfo, drop=_drop(p, seed, site))
    return H, X


@register_fake(f"{NS}::glove_project")
def _(ids, table, W, b, p, seed, site):
    return W.new_empty(ids.numel(), W.shape[0], dtype=torch.float32), W.new_empty(ids.numel(), table.shape[1], dtype=torch.float32)


@_cuda("glove_project_bwd")
def glove_project_bwd(gH: torch.Tensor, X: torch.Tensor, ids: torch.Tensor, W: torch.Tensor, p: float, seed: int,
                      site: int) -> Tuple[torch.Tensor, torch.Tensor]:
    D = W.shape[0]
    flat = ids.to(torch.int32).contiguous()
    rowinfo = torch.where(flat >= 0, torch.full_like(flat, 4), torch.zeros_like(flat))
    g = _f(gH).clone()
    call("lego_mask_dropout_rows", K._ptr(g), D, g.shape[0], None, D, K._ptr(rowinfo), K._drop(_drop(p, seed, site)), None, K._stream())
    gW = torch.zeros(W.shape, dtype=torch.float32, device=W.device)
    K.linear_bwd_weight(g, _f(X), gW)
    gb = torch.zeros(D, dtype=torch.float32, device=W.device)
    K.colsum(g, gb)
    return gW, gb


@register_fake(f"{NS}::glove_project_bwd")
def _(gH, X, ids, W, p, seed, site):
    return torch.empty_like(W, dtype=torch.float32), W.new_empty(W.shape[0], dtype=torch.float32)


@_cuda("glove_project_bwd_table")
def glove_project_bwd_table(gH: torch.Tensor, ids: torch.Tensor, W: torch.Tensor, rows: int, p: float, seed: int, site: int) -> torch.Tensor:
    """gradient of an UN-FROZEN pre-trained table under Transformation (`load_pretrained_embedding(..., frozen=False)`,
    loader/embedding_hub.py:171,262: nn.Embedding.from_pretrained with requires_grad switched on): the mask + Dropout backward of
    gH (the same Philox bits as the forward), dX = g W, rows added to the DENSE [rows, E0] gradient the reference's autograd
    produces (pad ids -1 contribute nothing: the reference's masked positions look up row 0 and are zeroed after the projection)"""
    D = W.shape[0]
    flat = ids.to(torch.int32).contiguous()
    rowinfo = torch.where(flat >= 0, torch.full_like(flat, 4), torch.zeros_like(flat))
    g = _f(gH).clone()
    call("lego_mask_dropout_rows", K._ptr(g), D, g.shape[0], None, D, K._ptr(rowinfo), K._drop(_drop(p, seed, site)), None, K._stream())
    gX = K.linear_bwd_data(g, _f(W))
    gT = torch.zeros(rows, W.shape[1], dtype=torch.float32, device=W.device)
    return K.scatter_add_rows(gT, flat, gX)


@register_fake(f"{NS}::glove_project_bwd_table")
def _(gH, ids, W, rows, p, seed, site):
    return W.new_empty(rows, W.shape[1], dtype=torch.float32)


def _gp_setup(ctx, inputs, output):
    ids, table, W, b, p, seed, site = inputs
    ctx.save_for_backward(output[1], ids, W)
    ctx.rng = (p, seed, site)
    ctx.table_rows = table.shape[0]


def _gp_bwd(ctx, gH, gX):
    X, ids, W = ctx.saved_tensors
    gW, gb = torch.ops.lego_hip.glove_project_bwd(gH.contiguous(), X, ids, W, *ctx.rng)
    gT = None
    if ctx.needs_input_grad[1]:                  # un-frozen pre-trained table
        gT = torch.ops.lego_hip.glove_project_bwd_table(gH.contiguous(), ids, W, ctx.table_rows, *ctx.rng)
    return None, gT, gW, gb, None, None, None


register_autograd(f"{NS}::glove_project", _gp_bwd, setup_context=_gp_setup)


# =============================================================================== CNNOperator title branch (a5)
@_cuda("conv3_relu_mask")
def conv3_relu_mask(h: torch.Tensor, mask: torch.Tensor, w: torch.Tensor, b: torch.Tensor, p: float, seed: int,
                    site: int) -> torch.Tensor:
    """Conv1d(k=3, 'same') -> ReLU -> * mask -> Dropout on [n,L,Din] (model/operators/cnn_operator.py:54-57)"""
    n, L, Din = h.shape
    plan = K.plan_dense(mask)
    y = K.conv3_fwd(_f(h).view(n * L, Din), K.conv3_pack(w), b, plan, drop=_drop(p, seed, site))
    return y.view(n, L, w.shape[0])


@register_fake(f"{NS}::conv3_relu_mask")
def _(h, mask, w, b, p, seed, site):
    return h.new_empty(h.shape[0], h.shape[1], w.shape[0], dtype=torch.float32)


@_cuda("conv3_relu_mask_bwd")
def conv3_relu_mask_bwd(gy: torch.Tensor, h: torch.Tensor, y: torch.Tensor, mask: torch.Tensor, w: torch.Tensor,
                        p: float) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor]:
    n, L, Din = h.shape
    Dout = w.shape[0]
    plan = K.plan_dense(mask)
    wt = K.conv3_pack(w)
    g = _f(gy).reshape(n * L, Dout).clone()
    scale = 1.0 / (1.0 - p) if p > 0.0 else 1.0          # y > 0 <=> kept and pre-activation > 0: the mask is in y itself
    y2 = _f(y).view(n * L, Dout)
    call("lego_relu_bwd", K._ptr(g), Dout, K._ptr(y2), Dout, n * L, Dout, float(scale), K._stream())
    gb = torch.zeros(Dout, dtype=torch.float32, device=g.device)
    K.colsum(g, gb)
    dwt = torch.zeros(3, Dout, Din, dtype=torch.float32, device=g.device)
    K.conv3_bwd_weight(g, _f(h).view(n * L, Din), plan, dwt)
    gw = torch.zeros(w.shape, dtype=torch.float32, device=g.device)
    K.conv3_unpack_add(dwt, gw)
    # d h: rows are NOT re-masked here (the live factor belongs to the producer of h)
    gh = K.conv3_bwd_data(g, wt, K.DensePlan(None, n, L, g.device), Din)
    return gh.view(n, L, Din), gw, gb


@register_fake(f"{NS}::conv3_relu_mask_bwd")
def _(gy, h, y, mask, w, p):
    return torch.empty_like(h, dtype=torch.float32), torch.empty_like(w, dtype=torch.float32), w.new_empty(w.shape[0], dtype=torch.float32)


def _conv_setup(ctx, inputs, output):
    h, mask, w, b, p, seed, site = inputs
    ctx.save_for_backward(h, output, mask, w)
    ctx.p = p


def _conv_bwd(ctx, gy):
    h, y, mask, w = ctx.saved_tensors
    gh, gw, gb = torch.ops.lego_hip.conv3_relu_mask_bwd(gy.contiguous(), h, y, mask, w, ctx.p)
    return gh, None, gw, gb, None, None, None


register_autograd(f"{NS}::conv3_relu_mask", _conv_bwd, setup_context=_conv_setup)


# =============================================================================== AdditiveAttention (a6 / a7)
@_cuda("additive_pool")
def additive_pool(x: torch.Tensor, mask: torch.Tensor, W1: torch.Tensor, b1: torch.Tensor,
                  w2: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor]:
    """AdditiveAttention.forward on [n,L,D] + mask [n,L] (model/common/attention.py:31-38): tanh hidden on the MFMA
    kernel, LDS-staged pool with wavefront-shuffle softmax reductions.  Returns (out [n,D], tanh hidden [n*L,A],
    attention weights [n*L]) -- the last two are what the backward needs."""
    out, c = K.additive_attention_fwd(x, mask, W1, b1, w2)
    return out, c.t, c.wrow


@register_fake(f"{NS}::additive_pool")
def _(x, mask, W1, b1, w2):
    n, L, D = x.shape
    return x.new_empty(n, D, dtype=torch.float32), x.new_empty(n * L, W1.shape[0], dtype=torch.float32), x.new_empty(n * L, dtype=torch.float32)


@_cuda("additive_pool_bwd")
def additive_pool_bwd(gout: torch.Tensor, x: torch.Tensor, mask: torch.Tensor, W1: torch.Tensor, w2: torch.Tensor,
                      t: torch.Tensor, wrow: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor, torch.Tensor]:
    n, L, D = x.shape
    c = K._AddCtx()
    c.x, c.t, c.wrow, c.plan = _f(x).view(n * L, D), t.clone(), wrow, K.plan_dense(mask)     # the pool backward overwrites t
    c.W1, c.w2, c.shape = _f(W1), _f(w2).view(-1), (n, L, D, W1.shape[0])
    gx, gW1, gb1, gw2 = K.additive_attention_bwd(c, _f(gout))
    return gx, gW1, gb1, gw2


@register_fake(f"{NS}::additive_pool_bwd")
def _(gout, x, mask, W1, w2, t, wrow):
    A = W1.shape[0]
    return (torch.empty_like(x, dtype=torch.float32), torch.empty_like(W1, dtype=torch.float32), W1.new_empty(A, dtype=torch.float32),
            W1.new_empty(1, A, dtype=torch.float32))


def _add_setup(ctx, inputs, output):
    x, mask, W1, b1, w2 = inputs
    ctx.save_for_backward(x, mask, W1, w2, output[1], output[2])
    ctx.w2_shape = w2.shape


def _add_bwd(ctx, gout, gt, gw):
    x, mask, W1, w2, t, wrow = ctx.saved_tensors
    gx, gW1, gb1, gw2 = torch.ops.lego_hip.additive_pool_bwd(gout.contiguous(), x, mask, W1, w2, t, wrow)
    return gx, None, gW1, gb1, gw2.view(ctx.w2_shape)


register_autograd(f"{NS}::additive_pool", _add_bwd, setup_context=_add_setup)


# =============================================================================== nn.MultiheadAttention (a8)
@_cuda("mhsa")
def mhsa(x: torch.Tensor, mask: torch.Tensor, in_w: torch.Tensor, in_b: torch.Tensor, out_w: torch.Tensor,
         out_b: torch.Tensor, heads: int, p: float, seed: int, site: int) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor, torch.Tensor, torch.Tensor]:
    """nn.MultiheadAttention(q=k=v=x, key_padding_mask=1-mask) (model/operators/attention_operator.py:46-50): MFMA in/out
    projections + the attention core.  Returns (y [n,L,D], compact x rows, qkv, head outputs, log-sum-exp of the score rows);
    the saved activations hold the R live rows first inside n*L-row buffers the kernels wrote in place, so every output shape
    is static and nothing is copied (the probabilities are not saved: the backward pass recomputes them)."""
    y, c = K.mhsa_fwd(x, mask, in_w, in_b, out_w, out_b, heads, drop=_drop(p, seed, site), padded=True)
    return (y,) + c.saved


@register_fake(f"{NS}::mhsa")
def _(x, mask, in_w, in_b, out_w, out_b, heads, p, seed, site):
    n, L, D = x.shape
    R = n * L
    f = dict(dtype=torch.float32)
    return (x.new_empty(n, L, D, **f), x.new_empty(R, D, **f), x.new_empty(R, 3 * D, **f), x.new_empty(R, D, **f),
            x.new_empty(R, heads, **f))


@_cuda("mhsa_bwd")
def mhsa_bwd(gy: torch.Tensor, mask: torch.Tensor, in_w: torch.Tensor, out_w: torch.Tensor, xc: torch.Tensor,
             qkv: torch.Tensor, o: torch.Tensor, lse: torch.Tensor, heads: int, p: float, seed: int,
             site: int) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor, torch.Tensor, torch.Tensor]:
    n, L, D = gy.shape
    c = K._MhsaCtx()
    c.idx, c.seg_off = K._compact(mask)
    R = c.idx.numel()
    c.xc, c.qkv, c.o, c.lse = xc[:R], qkv[:R], o[:R], lse[:R]
    c.in_w, c.out_w, c.heads, c.shape, c.drop = _f(in_w), _f(out_w), heads, (n, L, D), _drop(p, seed, site)
    return K.mhsa_bwd(c, _f(gy))


@register_fake(f"{NS}::mhsa_bwd")
def _(gy, mask, in_w, out_w, xc, qkv, o, lse, heads, p, seed, site):
    f = dict(dtype=torch.float32)
    D = gy.shape[2]
    return (torch.empty_like(gy, **f), torch.empty_like(in_w, **f), in_w.new_empty(3 * D, **f), torch.empty_like(out_w, **f),
            in_w.new_empty(D, **f))


def _mhsa_setup(ctx, inputs, output):
    x, mask, in_w, in_b, out_w, out_b, heads, p, seed, site = inputs
    ctx.save_for_backward(mask, in_w, out_w, *output[1:])
    ctx.args = (heads, p, seed, site)


def _mhsa_bwd(ctx, gy, *unused):
    mask, in_w, out_w, xc, qkv, o, lse = ctx.saved_tensors
    gx, gin_w, gin_b, gout_w, gout_b = torch.ops.lego_hip.mhsa_bwd(gy.contiguous(), mask, in_w, out_w, xc, qkv, o, lse, *ctx.args)
    return gx, None, gin_w, gin_b, gout_w, gout_b, None, None, None, None


register_autograd(f"{NS}::mhsa", _mhsa_bwd, setup_context=_mhsa_setup)


# =============================================================================== predictors / loss (a9 / a10)
@_cuda("rowdot")
def rowdot(u: torch.Tensor, it: torch.Tensor) -> torch.Tensor:
    """DotPredictor: sum(u * i, -1) on [n,D] pairs (model/predictors/dot_predictor.py:7-10)"""
    u2, i2 = _f(u), _f(it)
    n, D = u2.shape
    out = torch.empty(n, dtype=torch.float32, device=u.device)
    call("lego_rowdot_fwd", K._ptr(u2), D, K._ptr(i2), D, n, D, K._ptr(out), K._stream())
    return out


@register_fake(f"{NS}::rowdot")
def _(u, it):
    return u.new_empty(u.shape[0], dtype=torch.float32)


@_cuda("rowdot_bwd")
def rowdot_bwd(g: torch.Tensor, u: torch.Tensor, it: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
    u2, i2, g2 = _f(u), _f(it), _f(g)
    n, D = u2.shape
    gu, gi = torch.empty_like(u2), torch.empty_like(i2)
    call("lego_rowdot_bwd", K._ptr(u2), D, K._ptr(i2), D, K._ptr(g2), n, D, K._ptr(gu), D, K._ptr(gi), D, K._stream())
    return gu, gi


@register_fake(f"{NS}::rowdot_bwd")
def _(g, u, it):
    return torch.empty_like(u, dtype=torch.float32), torch.empty_like(it, dtype=torch.float32)


register_autograd(f"{NS}::rowdot", lambda ctx, g: torch.ops.lego_hip.rowdot_bwd(g.contiguous(), *ctx.saved_tensors),
                  setup_context=lambda ctx, inputs, output: ctx.save_for_backward(*inputs))


@_cuda("dot_ce")
def dot_ce(user: torch.Tensor, items: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
    """batched user x candidate scoring + CrossEntropy(label 0): scores[b,c] = <user[b], items[b,c]>, loss = mean_b
    (model/legommender.py:254,263 with the Dot predictor).  Returns (loss [], scores [B,C])."""
    scores, loss = K.dot_ce_fwd(user, items)
    return loss.view(()), scores


@register_fake(f"{NS}::dot_ce")
def _(user, items):
    return user.new_empty((), dtype=torch.float32), user.new_empty(items.shape[0], items.shape[1], dtype=torch.float32)


@_cuda("dot_ce_bwd")
def dot_ce_bwd(gloss: torch.Tensor, user: torch.Tensor, items: torch.Tensor, scores: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
    gu, gi = K.dot_ce_bwd(user, items, scores, gloss=1.0)
    s = gloss.to(torch.float32).reshape(())
    return gu * s, gi * s


@register_fake(f"{NS}::dot_ce_bwd")
def _(gloss, user, items, scores):
    return torch.empty_like(user, dtype=torch.float32), torch.empty_like(items, dtype=torch.float32)


def _dotce_setup(ctx, inputs, output):
    ctx.save_for_backward(inputs[0], inputs[1], output[1])
    ctx.set_materialize_grads(True)


def _dotce_bwd(ctx, gloss, gscores):
    user, items, scores = ctx.saved_tensors
    return torch.ops.lego_hip.dot_ce_bwd(gloss, user, items, scores)


register_autograd(f"{NS}::dot_ce", _dotce_bwd, setup_context=_dotce_setup)


# =============================================================================== optimiser / sampler (a11 / a13)
@_cuda("adam_step", mutates_args=("p", "g", "m", "v"))
def adam_step(p: torch.Tensor, g: torch.Tensor, m: torch.Tensor, v: torch.Tensor, lr: float, step: int, grad_scale: float,
              zero_grad: bool) -> None:
    """torch.optim.Adam defaults (base_lego.py:201-204) over one flat fp32 buffer, gradient scaled by grad_scale first
    (1/world after the all-reduce) and optionally cleared as it is consumed"""
    call("lego_adam_step", K._ptr(p), K._ptr(g), K._ptr(m), K._ptr(v), p.numel(), float(lr), 0.9, 0.999, 1e-8, int(step),
         float(grad_scale), 1 if zero_grad else 0, K._stream())


@_cuda("sample_negatives")
def sample_negatives(row_user: torch.Tensor, row_item: torch.Tensor, neg_list: torch.Tensor, neg_len: torch.Tensor, K_neg: int,
                     n_items: int, seed: int, step: int, row_base: int, row_stride: int) -> torch.Tensor:
    """Resampler.rebuild_candidates (loader/resampler.py:159-171) on device: [B, K+1] candidates, positive first"""
    B = row_user.numel()
    cand = torch.empty(B, K_neg + 1, dtype=torch.int32, device=row_user.device)
    call("lego_sample_negatives", K._ptr(row_user), K._ptr(row_item), K._ptr(neg_list), K._ptr(neg_len), neg_list.shape[1], B, K_neg,
         n_items, int(seed), int(step), int(row_base), int(row_stride), None, K._ptr(cand), K._stream())
    return cand


@register_fake(f"{NS}::sample_negatives")
def _(row_user, row_item, neg_list, neg_len, K_neg, n_items, seed, step, row_base, row_stride):
    return row_user.new_empty(row_user.numel(), K_neg + 1, dtype=torch.int32)


@_cuda("gather_history")
def gather_history(row_user: torch.Tensor, user_hist: torch.Tensor, user_hist_len: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
    """the DataSet row copy (loader/data_set.py:61-85): click history and its length of each row's user"""
    B, S = row_user.numel(), user_hist.shape[1]
    hist = torch.empty(B, S, dtype=torch.int32, device=row_user.device)
    hl = torch.empty(B, dtype=torch.int32, device=row_user.device)
    call("lego_gather_history", K._ptr(row_user), K._ptr(user_hist), K._ptr(user_hist_len), B, S, K._ptr(hist), K._ptr(hl), K._stream())
    return hist, hl


@register_fake(f"{NS}::gather_history")
def _(row_user, user_hist, user_hist_len):
    return row_user.new_empty(row_user.numel(), user_hist.shape[1], dtype=torch.int32), row_user.new_empty(row_user.numel(), dtype=torch.int32)


OPS = ("gather_rows", "scatter_add_rows", "linear", "linear_bwd", "glove_project", "glove_project_bwd", "glove_project_bwd_table", "conv3_relu_mask",
       "conv3_relu_mask_bwd", "additive_pool", "additive_pool_bwd", "mhsa", "mhsa_bwd", "rowdot", "rowdot_bwd", "dot_ce",
       "dot_ce_bwd", "adam_step", "sample_negatives", "gather_history")
