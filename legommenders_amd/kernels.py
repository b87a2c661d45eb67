"""Tensor-level wrappers of the C ABI (include/lego_hip.h) for dense, operator-shaped inputs.

These are what the plug-in operators (`model/operators/*`) and the parity tests call: they take
torch CUDA tensors in the reference's layouts ([n,L,D] embeddings + [n,L] masks), build a dense
row plan on the device and run the same kernels the ragged training engine uses.  torch is used
only to own memory / the stream and for index bookkeeping; all arithmetic is in liblego_hip.so.
"""
from __future__ import annotations

import ctypes
from typing import Optional, Tuple

import torch

from . import _lib
from ._lib import LegoDropout, call


def _ptr(t: Optional[torch.Tensor], off: int = 0):
    if t is None:
        return None
    return ctypes.c_void_p(t.data_ptr() + off * t.element_size())


def _stream():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def _f32(t: torch.Tensor, name="tensor") -> torch.Tensor:
    if not t.is_cuda:
        raise _lib.LegoHipError(f"{name} must be a CUDA tensor: the HIP path has no CPU fallback")
    return t.detach().to(torch.float32).contiguous()


def _drop(drop):
    if drop is None or drop[0] <= 0:
        return None
    return ctypes.byref(LegoDropout(float(drop[0]), int(drop[1]), int(drop[2])))


class DensePlan:
    """rowinfo / segment offsets of n sequences x L positions (lego_plan_dense)."""

    def __init__(self, mask: Optional[torch.Tensor], n: int, L: int, device):
        self.n, self.L = n, L
        self.counters = torch.zeros(8, dtype=torch.int32, device=device)
        self.seg_off = torch.zeros(n + 1, dtype=torch.int32, device=device)
        self.rowinfo = torch.zeros(n * L, dtype=torch.int32, device=device)
        m = None if mask is None else mask.to(torch.int32).contiguous()
        call("lego_plan_dense", _ptr(m), n, L, _ptr(self.counters), _ptr(self.seg_off), _ptr(self.rowinfo), _stream())


def plan_dense(mask: torch.Tensor) -> DensePlan:
    n, L = mask.shape
    return DensePlan(mask, n, L, mask.device)


# --------------------------------------------------------------------------- linear family
def linear_fwd(x, W, b=None, act=0, rowinfo=None, drop=None, out=None):
    x, W = _f32(x, "x"), _f32(W, "W")
    M, K = x.shape
    N = W.shape[0]
    y = out if out is not None else torch.empty(M, N, dtype=torch.float32, device=x.device)
    call("lego_linear_fwd", _ptr(x), K, _ptr(W), K, _ptr(None if b is None else _f32(b)), _ptr(y), N, M, None, N, K, act,
         _ptr(rowinfo), _drop(drop), None, None, _stream())
    return y


def linear_bwd_data(g, W, accumulate_into=None, relu_ref=None, relu_scale=1.0, rowinfo=None, drop=None, colsum=None):
    g, W = _f32(g, "g"), _f32(W, "W")
    M, N = g.shape
    K = W.shape[1]
    dx = accumulate_into if accumulate_into is not None else torch.empty(M, K, dtype=torch.float32, device=g.device)
    call("lego_linear_bwd_data", _ptr(g), N, _ptr(W), K, _ptr(dx), K, M, None, N, K, 1 if accumulate_into is not None else 0,
         _ptr(relu_ref), K, float(relu_scale), _ptr(rowinfo), _drop(drop), _ptr(colsum), None, None, _stream())
    return dx


def linear_bwd_weight(g, x, dW):
    g, x = _f32(g, "g"), _f32(x, "x")
    M, N = g.shape
    K = x.shape[1]
    call("lego_linear_bwd_weight", _ptr(g), N, _ptr(x), K, _ptr(dW), K, M, None, N, K, None, None, _stream())
    return dW


def colsum(x, out):
    x = _f32(x)
    call("lego_colsum", _ptr(x), x.shape[1], x.shape[0], None, None, x.shape[1], _ptr(out), _stream())
    return out


# --------------------------------------------------------------------------- conv (k=3, 'same')
def conv3_pack(w):
    w = _f32(w, "cnn.weight")
    Dout, Din, k = w.shape
    assert k == 3, "the HIP conv kernel implements kernel_size=3 (config/model/naml.yaml:14)"
    wt = torch.empty(3, Dout, Din, dtype=torch.float32, device=w.device)
    call("lego_conv3_pack", _ptr(w), _ptr(wt), Dout, Din, _stream())
    return wt


def conv3_unpack_add(dwt, dw):
    _, Dout, Din = dwt.shape
    call("lego_conv3_unpack_add", _ptr(dwt), _ptr(dw), Dout, Din, _stream())
    return dw


def conv3_fwd(h, wt, b, plan: DensePlan, drop=None):
    h = _f32(h, "h")
    R, Din = h.shape
    Dout = wt.shape[1]
    y = torch.empty(R, Dout, dtype=torch.float32, device=h.device)
    call("lego_conv3_fwd", _ptr(h), Din, _ptr(wt), _ptr(_f32(b)), _ptr(plan.rowinfo), _ptr(y), Dout, R, None, Dout, Din,
         _drop(drop), 1, _stream())
    return y


def conv3_bwd_data(gy, wt, plan: DensePlan, Din, drop_in=None, colsum_out=None):
    gy = _f32(gy, "gy")
    R, Dout = gy.shape
    dh = torch.empty(R, Din, dtype=torch.float32, device=gy.device)
    call("lego_conv3_bwd_data", _ptr(gy), Dout, _ptr(wt), _ptr(plan.rowinfo), _ptr(dh), Din, R, None, Dout, Din,
         _drop(drop_in), _ptr(colsum_out), 1, _stream())
    return dh


def conv3_bwd_weight(gy, h, plan: DensePlan, dwt):
    gy, h = _f32(gy), _f32(h)
    R, Dout = gy.shape
    Din = h.shape[1]
    call("lego_conv3_bwd_weight", _ptr(gy), Dout, _ptr(h), Din, _ptr(plan.rowinfo), _ptr(dwt), R, None, Dout, Din, _stream())
    return dwt


# --------------------------------------------------------------------------- additive attention
class _AddCtx:
    pass


def additive_attention_fwd(x, mask, W1, b1, w2, extra: Optional[torch.Tensor] = None):
    """AdditiveAttention.forward on dense [n,L,D] + mask[n,L]; returns ([n,D], ctx)."""
    x = _f32(x, "x")
    n, L, D = x.shape
    A = W1.shape[0]
    plan = plan_dense(mask)
    xf = x.view(n * L, D)
    t = linear_fwd(xf, W1, b1, act=2)
    out = torch.empty(n, D, dtype=torch.float32, device=x.device)
    wrow = torch.empty(n * L, dtype=torch.float32, device=x.device)
    w2 = _f32(w2).view(-1)
    call("lego_additive_pool_fwd", _ptr(t), A, _ptr(xf), D, _ptr(w2), _ptr(plan.seg_off), _ptr(plan.rowinfo), None,
         n, None, D, A, _ptr(out), D, _ptr(wrow), _stream())
    ctx = _AddCtx()
    ctx.x, ctx.t, ctx.wrow, ctx.plan, ctx.W1, ctx.w2, ctx.shape = xf, t, wrow, plan, _f32(W1), w2, (n, L, D, A)
    return out, ctx


def additive_attention_bwd(ctx, gout):
    n, L, D, A = ctx.shape
    gout = _f32(gout)
    dev = gout.device
    dx = torch.empty(n * L, D, dtype=torch.float32, device=dev)
    gw2 = torch.zeros(A, dtype=torch.float32, device=dev)
    gb1 = torch.zeros(A, dtype=torch.float32, device=dev)
    gW1 = torch.zeros(A, D, dtype=torch.float32, device=dev)
    call("lego_additive_pool_bwd", _ptr(ctx.t), A, _ptr(ctx.x), D, _ptr(ctx.w2), _ptr(ctx.plan.seg_off), None, n, None,
         D, A, _ptr(gout), D, _ptr(ctx.wrow), _ptr(dx), D, _ptr(gw2), _ptr(gb1), None, _stream())
    linear_bwd_weight(ctx.t, ctx.x, gW1)
    linear_bwd_data(ctx.t, ctx.W1, accumulate_into=dx)
    return dx.view(n, L, D), gW1, gb1, gw2.view(1, A)


# --------------------------------------------------------------------------- dot + CE
def dot_ce_fwd(user, items, with_loss=True):
    user, items = _f32(user), _f32(items)
    B, D = user.shape
    C = items.shape[1] if items.dim() == 3 else items.shape[0] // B
    it = items.reshape(B * C, D)
    scores = torch.empty(B, C, dtype=torch.float32, device=user.device)
    loss = torch.zeros(1, dtype=torch.float32, device=user.device)
    call("lego_dot_ce_fwd", _ptr(user), D, _ptr(it), D, B, C, D, _ptr(scores), _ptr(loss) if with_loss else None, _stream())
    return scores, loss


def dot_ce_bwd(user, items, scores, gloss=1.0):
    user, items = _f32(user), _f32(items)
    B, D = user.shape
    C = scores.shape[1]
    it = items.reshape(B * C, D)
    gu = torch.empty_like(user)
    gi = torch.empty_like(it)
    call("lego_dot_ce_bwd", _ptr(user), D, _ptr(it), D, _ptr(scores), B, C, D, float(gloss) / B, None, _ptr(gu), D, _ptr(gi), D, _stream())
    return gu, gi.view(B, C, D)


def adam_step(p, g, m, v, lr, step, betas=(0.9, 0.999), eps=1e-8, grad_scale=1.0):
    n = p.numel()
    call("lego_adam_step", _ptr(p), _ptr(g), _ptr(m), _ptr(v), n, float(lr), float(betas[0]), float(betas[1]), float(eps),
         int(step), float(grad_scale), 0, _stream())


# --------------------------------------------------------------------------- MHSA (dense API, any 0/1 mask)
class _MhsaCtx:
    pass


def _compact(mask):
    """live-row index of a [n,L] mask and the ragged segment offsets (index bookkeeping only)."""
    n, L = mask.shape
    live = mask.reshape(-1) != 0
    idx = torch.nonzero(live, as_tuple=False).view(-1).to(torch.int32)
    lens = mask.to(torch.int64).sum(1)
    seg_off = torch.zeros(n + 1, dtype=torch.int32, device=mask.device)
    seg_off[1:] = torch.cumsum(lens, 0).to(torch.int32)
    return idx, seg_off


def gather_rows(table, idx, out=None):
    table = _f32(table)
    R = idx.numel()
    W = table.shape[1]
    out = out if out is not None else torch.empty(R, W, dtype=torch.float32, device=table.device)
    call("lego_gather_rows", _ptr(table), W, W, _ptr(idx), R, None, _ptr(out), W, 0, _stream())
    return out


def scatter_add_rows(grad_table, idx, g):
    g = _f32(g)
    W = g.shape[1]
    call("lego_scatter_add_rows", _ptr(grad_table), grad_table.shape[1], W, grad_table.shape[0], _ptr(idx), idx.numel(), None,
         _ptr(g), W, _stream())
    return grad_table


def mhsa_fwd(x, mask, in_w, in_b, out_w, out_b, heads, drop=None, padded=False):
    """nn.MultiheadAttention(q=k=v=x, key_padding_mask=(1-mask)) on dense [n,L,D]; rows whose mask is 0
    are neither keys nor queries (their output rows are returned as zeros).
    `padded`: the saved activations (compact x rows, qkv, head outputs, log-sum-exp) are allocated at n*L rows with the R live
    rows first and zeros behind -- static shapes for the dispatcher route, written in place (no copy into a padded tensor)."""
    x = _f32(x)
    n, L, D = x.shape
    idx, seg_off = _compact(mask)
    R = idx.numel()
    cap = n * L if padded else R
    alloc = torch.zeros if padded else torch.empty
    f = dict(dtype=torch.float32, device=x.device)
    xc_, qkv_, o_, lse_ = alloc(cap, D, **f), alloc(cap, 3 * D, **f), alloc(cap, D, **f), alloc(cap, heads, **f)
    xc = gather_rows(x.view(n * L, D), idx, out=xc_[:R])
    qkv = linear_fwd(xc, in_w, in_b, out=qkv_[:R])
    o, lse = o_[:R], lse_[:R]
    if R > 0:                                       # an all-masked batch has no rows: lse[:0] is a NULL pointer, which the entry point refuses
        call("lego_mhsa_core_fwd", _ptr(qkv), 3 * D, _ptr(seg_off), n, None, D, heads, _ptr(o), D, _ptr(lse), None, L,
             _drop(drop), R, 0, None, None, _stream())
    yc = linear_fwd(o, out_w, out_b)
    y = torch.zeros(n * L, D, dtype=torch.float32, device=x.device)
    y.index_copy_(0, idx.long(), yc)
    ctx = _MhsaCtx()
    ctx.xc, ctx.qkv, ctx.o, ctx.lse, ctx.idx, ctx.seg_off = xc, qkv, o, lse, idx, seg_off
    ctx.saved = (xc_, qkv_, o_, lse_)
    ctx.in_w, ctx.out_w, ctx.heads, ctx.shape, ctx.drop = _f32(in_w), _f32(out_w), heads, (n, L, D), drop
    return y.view(n, L, D), ctx


def mhsa_bwd(ctx, gy):
    n, L, D = ctx.shape
    dev = gy.device
    R = ctx.idx.numel()
    gyc = gather_rows(_f32(gy).view(n * L, D), ctx.idx)
    gout_w = torch.zeros(D, D, dtype=torch.float32, device=dev)
    gout_b = torch.zeros(D, dtype=torch.float32, device=dev)
    linear_bwd_weight(gyc, ctx.o, gout_w)
    colsum(gyc, gout_b)
    go = linear_bwd_data(gyc, ctx.out_w)
    gqkv = torch.empty(R, 3 * D, dtype=torch.float32, device=dev)
    if R > 0:
        call("lego_mhsa_core_bwd", _ptr(ctx.qkv), 3 * D, _ptr(ctx.seg_off), n, None, D, ctx.heads, _ptr(go), D, _ptr(ctx.lse), None, L,
             _drop(ctx.drop), R, _ptr(gqkv), 3 * D, None, 0, None, None, _stream())
    gin_w = torch.zeros(3 * D, D, dtype=torch.float32, device=dev)
    gin_b = torch.zeros(3 * D, dtype=torch.float32, device=dev)
    linear_bwd_weight(gqkv, ctx.xc, gin_w)
    colsum(gqkv, gin_b)
    gxc = linear_bwd_data(gqkv, ctx.in_w)
    gx = torch.zeros(n * L, D, dtype=torch.float32, device=dev)
    gx.index_copy_(0, ctx.idx.long(), gxc)
    return gx.view(n, L, D), gin_w, gin_b, gout_w, gout_b
