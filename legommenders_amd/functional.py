"""torch.autograd Functions over the HIP kernels, in the reference's dense operator shapes.

This is the layer the plug-in classes (`model/operators/*`, `model/predictors/*`, `loader/embedding_hub`)
call, so third-party operators written against the reference's interface keep working while every
arithmetic step runs in liblego_hip.so.  The training fast path (`engine.py`) bypasses autograd entirely;
these Functions serve the operator-level API and its parity tests.  No CPU fallback: CPU tensors raise.
"""
from __future__ import annotations

import itertools

import torch
from torch.autograd import Function

from . import kernels as K
from ._lib import LegoHipError, call

_site = itertools.count(1)
SEED = 2023


def seed_streams(seed: int):
    """seed of the Philox dropout streams of this process (data parallel: train_step.rank_seed(seed, rank))"""
    global SEED
    SEED = int(seed)


def _drop(p, training):
    """(p, seed, site) with a fresh Philox stream per call, or None in eval / p == 0."""
    if not training or p <= 0:
        return None
    return (float(p), SEED, 1000 + next(_site))


def _need_gpu(t, what):
    if not t.is_cuda:
        raise LegoHipError(f"{what}: CPU tensors are not supported -- the MI355X path has no CPU fallback "
                           "(use the reference for --cuda -1)")


class _Linear(Function):
    @staticmethod
    def forward(ctx, x, W, b, act):
        _need_gpu(x, "linear")
        shape = x.shape
        x2 = K._f32(x).reshape(-1, shape[-1])
        if act != 0:
            raise LegoHipError("functional.linear implements nn.Linear only; tanh is fused inside additive_attention")
        y = K.linear_fwd(x2, W, b, act=0)
        ctx.save_for_backward(x2, K._f32(W))
        ctx.shape, ctx.has_b = shape, b is not None
        return y.view(*shape[:-1], W.shape[0])

    @staticmethod
    def backward(ctx, gy):
        x2, W = ctx.saved_tensors
        g = K._f32(gy).reshape(-1, W.shape[0])
        gW = torch.zeros_like(W)
        K.linear_bwd_weight(g, x2, gW)
        gb = None
        if ctx.has_b:
            gb = torch.zeros(W.shape[0], dtype=torch.float32, device=W.device)
            K.colsum(g, gb)
        gx = K.linear_bwd_data(g, W).view(ctx.shape)
        return gx, gW, gb, None


def linear(x, W, b=None, act=0):
    return _Linear.apply(x, W, b, act)


class _GloveProject(Function):
    """Transformation.forward + SimpleInputer masking: Dropout(Linear(Embedding(ids))) * mask
    (loader/embedding_hub.py:95-96, model/inputer/simple_inputer.py:55-63); frozen table."""

    @staticmethod
    def forward(ctx, ids, table, W, b, p, training):
        _need_gpu(ids, "embedding")
        shape = ids.shape
        flat = ids.reshape(-1).to(torch.int32).contiguous()            # pad ids are -1 -> zero rows, never looked up
        X = K.gather_rows(table, flat)
        rowinfo = torch.where(flat >= 0, torch.full_like(flat, 4), torch.zeros_like(flat))   # live bit only
        drop = _drop(p, training)
        H = K.linear_fwd(X, W, b, act=0, rowinfo=rowinfo, drop=drop)
        ctx.save_for_backward(X, rowinfo, K._f32(W))
        ctx.drop, ctx.shape = drop, shape
        return H.view(*shape, W.shape[0])

    @staticmethod
    def backward(ctx, gH):
        X, rowinfo, W = ctx.saved_tensors
        D = W.shape[0]
        g = K._f32(gH).reshape(-1, D).clone()
        call("lego_mask_dropout_rows", K._ptr(g), D, g.shape[0], None, D, K._ptr(rowinfo), K._drop(ctx.drop), K._stream())
        gW = torch.zeros_like(W)
        K.linear_bwd_weight(g, X, gW)
        gb = torch.zeros(D, dtype=torch.float32, device=W.device)
        K.colsum(g, gb)
        return None, None, gW, gb, None, None


def glove_project(ids, table, W, b, p=0.0, training=False):
    return _GloveProject.apply(ids, table, W, b, p, training)


class _Embedding(Function):
    """Trainable nn.Embedding look-up with the inputer's pad handling (id -1 -> zero row) and the
    reference's DENSE gradient semantics (embedding_hub.py:325-335)."""

    @staticmethod
    def forward(ctx, ids, table):
        _need_gpu(ids, "embedding")
        flat = ids.reshape(-1).to(torch.int32).contiguous()
        out = K.gather_rows(table, flat)
        ctx.save_for_backward(flat)
        ctx.tshape, ctx.shape = table.shape, ids.shape
        return out.view(*ids.shape, table.shape[1])

    @staticmethod
    def backward(ctx, g):
        (flat,) = ctx.saved_tensors
        gt = torch.zeros(ctx.tshape, dtype=torch.float32, device=g.device)
        K.scatter_add_rows(gt, flat, K._f32(g).reshape(-1, ctx.tshape[1]))
        return None, gt


def embedding(ids, table):
    return _Embedding.apply(ids, table)


class _Conv3ReluMask(Function):
    """CNNOperator title branch: Conv1d(k=3,'same') -> ReLU -> *mask -> Dropout (cnn_operator.py:54-57)."""

    @staticmethod
    def forward(ctx, h, mask, w, b, p, training):
        _need_gpu(h, "conv3")
        n, L, Din = h.shape
        plan = K.plan_dense(mask)
        wt = K.conv3_pack(w)
        drop = _drop(p, training)
        h2 = K._f32(h).view(n * L, Din)
        y = K.conv3_fwd(h2, wt, b, plan, drop=drop)
        ctx.save_for_backward(h2, y, wt)
        ctx.plan, ctx.scale, ctx.shape, ctx.wshape = plan, (1.0 / (1.0 - p) if drop else 1.0), (n, L, Din), w.shape
        return y.view(n, L, w.shape[0])

    @staticmethod
    def backward(ctx, gy):
        h2, y, wt = ctx.saved_tensors
        n, L, Din = ctx.shape
        Dout = ctx.wshape[0]
        g = K._f32(gy).reshape(n * L, Dout).clone()
        call("lego_relu_bwd", K._ptr(g), Dout, K._ptr(y), Dout, n * L, Dout, float(ctx.scale), K._stream())
        gb = torch.zeros(Dout, dtype=torch.float32, device=g.device)
        K.colsum(g, gb)
        dwt = torch.zeros(3, Dout, Din, dtype=torch.float32, device=g.device)
        K.conv3_bwd_weight(g, h2, ctx.plan, dwt)
        gw = torch.zeros(ctx.wshape, dtype=torch.float32, device=g.device)
        K.conv3_unpack_add(dwt, gw)
        # d h: rows are NOT re-masked here (the live factor belongs to the producer of h)
        all_live = K.DensePlan(None, n, L, g.device)
        gh = K.conv3_bwd_data(g, wt, all_live, Din)
        return gh.view(n, L, Din), None, gw, gb, None, None


def conv3_relu_mask(h, mask, w, b, p=0.0, training=False):
    return _Conv3ReluMask.apply(h, mask, w, b, p, training)


class _Additive(Function):
    @staticmethod
    def forward(ctx, x, mask, W1, b1, w2):
        _need_gpu(x, "additive_attention")
        y, c = K.additive_attention_fwd(x, mask, W1, b1, w2)
        ctx.c = c
        return y

    @staticmethod
    def backward(ctx, gy):
        gx, gW1, gb1, gw2 = K.additive_attention_bwd(ctx.c, gy.contiguous())
        return gx, None, gW1, gb1, gw2


def additive_attention(x, mask, W1, b1, w2):
    return _Additive.apply(x, mask, W1, b1, w2)


class _Mhsa(Function):
    @staticmethod
    def forward(ctx, x, mask, in_w, in_b, out_w, out_b, heads, p, training):
        _need_gpu(x, "multi_head_attention")
        y, c = K.mhsa_fwd(x, mask, in_w, in_b, out_w, out_b, heads, drop=_drop(p, training))
        ctx.c = c
        return y

    @staticmethod
    def backward(ctx, gy):
        gx, gin_w, gin_b, gout_w, gout_b = K.mhsa_bwd(ctx.c, gy.contiguous())
        return gx, None, gin_w, gin_b, gout_w, gout_b, None, None, None


def multi_head_self_attention(x, mask, in_w, in_b, out_w, out_b, heads, p=0.0, training=False):
    return _Mhsa.apply(x, mask, in_w, in_b, out_w, out_b, heads, p, training)


class _RowDot(Function):
    @staticmethod
    def forward(ctx, u, it):
        _need_gpu(u, "dot predictor")
        u2, i2 = K._f32(u), K._f32(it)
        n, D = u2.shape
        out = torch.empty(n, dtype=torch.float32, device=u.device)
        call("lego_rowdot_fwd", K._ptr(u2), D, K._ptr(i2), D, n, D, K._ptr(out), K._stream())
        ctx.save_for_backward(u2, i2)
        return out

    @staticmethod
    def backward(ctx, g):
        u2, i2 = ctx.saved_tensors
        n, D = u2.shape
        gu, gi = torch.empty_like(u2), torch.empty_like(i2)
        call("lego_rowdot_bwd", K._ptr(u2), D, K._ptr(i2), D, K._ptr(K._f32(g)), n, D, K._ptr(gu), D, K._ptr(gi), D, K._stream())
        return gu, gi


def rowdot(u, it):
    return _RowDot.apply(u, it)


class _DotCE(Function):
    """Fused DotPredictor + CrossEntropy(label 0) on [B,D] users x [B,C,D] candidates."""

    @staticmethod
    def forward(ctx, user, items):
        _need_gpu(user, "dot_ce")
        scores, loss = K.dot_ce_fwd(user, items)
        ctx.save_for_backward(K._f32(user), K._f32(items), scores)
        return loss.view(()), scores

    @staticmethod
    def backward(ctx, gloss, gscores):
        user, items, scores = ctx.saved_tensors
        gu, gi = K.dot_ce_bwd(user, items, scores, gloss=float(gloss))
        return gu, gi


def dot_ce(user, items):
    return _DotCE.apply(user, items)
