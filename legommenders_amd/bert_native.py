"""The BERT news encoder's transformer blocks on the path's own kernels (config 5).

The reference runs `transformers.BertModel(inputs_embeds=..., attention_mask=...)` (model/operators/once_operator.py:156-170,
bert_operator.py:16) or, in cached-layer mode, `BertEncoder(hidden_states=..., attention_mask=...)` on the kept blocks
(bert_operator.py:30-45).  Here the same function runs as ONE autograd node over RAGGED rows -- only the live positions of every
item are rows, nothing is padded to the page's longest sequence -- built from:

    products        lego_linear_fwd / _bwd_data / _bwd_weight   (fp32 MFMA row-strip and TN kernels; q / k / v write the three
                                                                 thirds of one [rows, 3H] buffer, the layout the attention core reads)
    attention       lego_mhsa_core_fwd / _bwd                    (head dim 64 instantiation, probability dropout in the kernel)
    LayerNorm tails lego_dropout_add_layernorm_fwd / _bwd        (dense output -> Dropout -> + residual -> LayerNorm in one pass)
    GELU            lego_gelu_fwd / _bwd

The HF module tree stays what it is (parameter names, state_dict keys, optimizer groups); this module only reads its parameter
tensors.  Published algorithm of the third-party package (transformers modeling_bert.py: BertEmbeddings, BertSelfAttention,
BertSelfOutput, BertIntermediate, BertOutput; pinned through the reference-generated fixtures tests/golden/bert_naml_*.npz and the
test suite's CPU restatement of a BERT block).  Measured against the HF route: DESIGN.md section 5 (config 5 table).

Where PyTorch-ROCm's hipBLASLt is used instead of the path's kernel: nowhere.  `tools/bert_shapes_bench.py` has the per-shape
comparison (forward / data-gradient products: hipBLASLt 0-15 % faster at the FFN shapes; weight gradients: the path's TN kernel
1.6x faster); the round-4 A/B hook that routed the FFN products through torch was removed in round 6.
"""
from __future__ import annotations

import ctypes
import os
from typing import List, Optional

import torch

from . import functional as F_hip
from . import kernels as K
from ._lib import LegoDropout, call
from .arena import arena_of
from .kernels import _ptr, _stream

LAYER_KEYS = ("attention.self.query.weight", "attention.self.query.bias", "attention.self.key.weight", "attention.self.key.bias",
              "attention.self.value.weight", "attention.self.value.bias", "attention.output.dense.weight", "attention.output.dense.bias",
              "attention.output.LayerNorm.weight", "attention.output.LayerNorm.bias", "intermediate.dense.weight", "intermediate.dense.bias",
              "output.dense.weight", "output.dense.bias", "output.LayerNorm.weight", "output.LayerNorm.bias")
EMBED_KEYS = ("position_embeddings.weight", "token_type_embeddings.weight", "LayerNorm.weight", "LayerNorm.bias")
NL = len(LAYER_KEYS)
# PluginStep (plugin_step.py) keeps every trainable parameter's gradient as a view of ONE flat buffer that its Adam launch reads and clears.
# With DIRECT_GRADS on, the backward pass below accumulates each parameter gradient straight into `param.grad` when that is such a pre-set
# fp32 tensor and hands autograd `None` for it: no per-step gradient buffer, no AccumulateGrad pass over 79 M elements.  Off (the default
# outside PluginStep): gradients are returned to autograd as usual.
DIRECT_GRADS = False
ROWS_SEEN = None        # a list: the live-row count of every forward pass is appended (tools/bert_naml_bench.py: step time per row)


def _stacked(ts):
    """ONE [sum of rows, ...] view over tensors that lie back to back in one storage (PluginStep's flat buffers put q, k, v so), else None"""
    t0 = ts[0]
    if any(t is None for t in ts):
        return None
    base, off = t0.untyped_storage().data_ptr(), t0.storage_offset()
    for t in ts:
        if (not t.is_contiguous() or t.dtype != t0.dtype or t.untyped_storage().data_ptr() != base or t.storage_offset() != off
                or t.shape[1:] != t0.shape[1:]):
            return None
        off += t.numel()
    return torch.as_strided(t0, (sum(t.shape[0] for t in ts),) + tuple(t0.shape[1:]), t0.stride())


def supported(transformer, L: int) -> Optional[str]:
    """None when the native blocks cover this transformer at sequence width L, else the reason (the caller falls back to HF)"""
    cfg = transformer.config
    H, heads = cfg.hidden_size, cfg.num_attention_heads
    if H % heads or (H // heads) not in (8, 16, 32, 64):
        return f"head dim {H}/{heads} not in (8, 16, 32, 64)"
    if L > 64:
        return f"sequence width {L} > 64 (the attention core's tile)"
    if getattr(cfg, "hidden_act", "gelu") != "gelu":
        return f"hidden_act {cfg.hidden_act!r} (exact GELU only)"
    if getattr(cfg, "position_embedding_type", "absolute") not in ("absolute", None):
        return "relative position embeddings"
    if H % 4 or cfg.intermediate_size % 4 or H > 1024:
        return "widths must be multiples of 4 and the hidden size <= 1024"
    layers = transformer.encoder.layer
    if len(layers) and not isinstance(layers[0].attention.self.query, torch.nn.Linear):
        return "adapter-wrapped projections (LoRA) run through the module tree"
    return None


def layer_params(transformer) -> List[torch.Tensor]:
    """the kept blocks' parameters in LAYER_KEYS order, block after block (live tensors of the HF modules)"""
    out = []
    for block in transformer.encoder.layer:
        sd = dict(block.named_parameters())
        out += [sd[k] for k in LAYER_KEYS]
    return out


def embed_params(transformer) -> List[torch.Tensor]:
    sd = dict(transformer.embeddings.named_parameters())
    return [sd[k] for k in EMBED_KEYS]


def _drop(p, training):
    """(ctypes byref or None, (p, seed, site)): a fresh Philox stream of the process-wide dropout seed per site"""
    p, seed, site = F_hip._rng(float(p), training)
    if p <= 0.0:
        return None, None
    return ctypes.byref(LegoDropout(p, int(seed), int(site))), (p, int(seed), int(site))


def _redrop(rng):
    return None if rng is None else ctypes.byref(LegoDropout(*rng))


# bench.py / tools/bert_naml_bench.py set TIMERS to a dict: every product launch is then bracketed by HIP events on the launch stream
# and recorded under its tag with its algorithmic flops (tag -> [(event0, event1, flops)]); None = no events (the normal path)
TIMERS = None


class _timed:
    def __init__(self, tag, flops):
        self.on = TIMERS is not None and tag is not None
        self.tag, self.flops = tag, flops

    def __enter__(self):
        if self.on:
            self.e0, self.e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            self.e0.record()

    def __exit__(self, *exc):
        if self.on:
            self.e1.record()
            TIMERS.setdefault(self.tag, []).append((self.e0, self.e1, self.flops))


def _lin_fwd(x, W, b, out, ldo, col=0, tag=None):
    """out[:, col : col + N] = x W^T + b  (out has leading dimension ldo)"""
    M, Kd = x.shape
    N = W.shape[0]
    with _timed(tag, 2.0 * M * N * Kd):
        call("lego_linear_fwd", _ptr(x), Kd, _ptr(W), Kd, _ptr(b), _ptr(out, col), ldo, M, None, N, Kd, 0, None, None, None, None, _stream())


def _lin_bwd_data(g, ldg, col, W, dx, accumulate, tag=None):
    """dx (+)= g[:, col : col + N] W   (g has leading dimension ldg)"""
    M = dx.shape[0]
    N, Kd = W.shape
    with _timed(tag, 2.0 * M * N * Kd):
        call("lego_linear_bwd_data", _ptr(g, col), ldg, _ptr(W), Kd, _ptr(dx), Kd, M, None, N, Kd, 1 if accumulate else 0,
             None, 0, 1.0, None, None, None, None, None, _stream())


def _lin_bwd_weight(g, ldg, col, x, gW, tag=None):
    """gW += g[:, col : col + N]^T x   (outputs of more than 1 M elements: the library issues them as 768-wide chunks, csrc/gemm_ops.hip)"""
    M, Kd = x.shape
    N = gW.shape[0]
    with _timed(tag, 2.0 * M * N * Kd):
        call("lego_linear_bwd_weight", _ptr(g, col), ldg, _ptr(x), Kd, _ptr(gW), gW.stride(0), M, None, N, Kd, None, None, _stream())


FUSED_GELU = True       # the GELU inside the two feed-forward products' epilogues (False: lego_gelu_fwd / _bwd as passes of their own -- the cross-check)


def _ffn1_fwd(a, W1, b1, z, g):
    """z = a W1^T + b1 (kept for GELU'), g = gelu(z)"""
    M, Kd = a.shape
    N = W1.shape[0]
    if FUSED_GELU:
        with _timed("ffn1_fwd", 2.0 * M * N * Kd):
            call("lego_linear_gelu_fwd", _ptr(a), Kd, _ptr(W1), Kd, _ptr(b1), _ptr(z), N, _ptr(g), N, M, N, Kd, _stream())
        return
    _lin_fwd(a, W1, b1, z, N, tag="ffn1_fwd")
    call("lego_gelu_fwd", _ptr(z), _ptr(g), z.numel(), _stream())


def _ffn2_bwd_data(d_fo, W2, z, dz):
    """dz = (d_fo W2) . gelu'(z)"""
    M, N = d_fo.shape
    Kd = W2.shape[1]
    if FUSED_GELU:
        with _timed("ffn2_bwd_data", 2.0 * M * N * Kd):
            call("lego_linear_bwd_data_gelu", _ptr(d_fo), N, _ptr(W2), Kd, _ptr(z), Kd, _ptr(dz), Kd, M, N, Kd, _stream())
        return
    _lin_bwd_data(d_fo, N, 0, W2, dz, False, tag="ffn2_bwd_data")
    call("lego_gelu_bwd", _ptr(dz), _ptr(z), _ptr(dz), dz.numel(), _stream())


def _colsum(g, ldg, col, N, out):
    call("lego_colsum", _ptr(g, col), ldg, g.shape[0], None, None, N, _ptr(out), _stream())


def _ln_fwd(y, resid, gamma, beta, eps, pre, post, out, mean, rstd):
    R, W = y.shape
    call("lego_dropout_add_layernorm_fwd", _ptr(y), W, _ptr(resid), W, _ptr(gamma), _ptr(beta), float(eps), pre, post, _ptr(out), W,
         _ptr(mean), _ptr(rstd), R, W, _stream())


def _ln_bwd(dout, y, resid, gamma, mean, rstd, pre, post, dy, dresid, dgamma, dbeta, dybias=None):
    R, W = y.shape
    call("lego_dropout_add_layernorm_bwd", _ptr(dout), W, _ptr(y), W, _ptr(resid), W, _ptr(gamma), _ptr(mean), _ptr(rstd), pre, post,
         _ptr(dy), W, _ptr(dresid), W, _ptr(dgamma), _ptr(dbeta), _ptr(dybias), R, W, _stream())


class _Blocks(torch.autograd.Function):
    """hidden states of the live rows through (optionally the embedding stage and) the kept blocks; dense in, dense out.
    Every tensor of the pass that does not leave it -- saved activations, LayerNorm statistics, backward temporaries -- is a view of the
    device's workspace arena (arena.py): a steady-state training step allocates nothing here."""

    @staticmethod
    def forward(ctx, x, mask, heads, eps, p_hidden, p_attn, training, embed, *params):
        n, L, H = x.shape
        dev = x.device
        x = x.detach().float().contiguous()
        idx, seg_off = K._compact(mask)
        R = int(idx.numel())
        f = dict(dtype=torch.float32, device=dev)
        n_layers = (len(params) - (len(EMBED_KEYS) if embed else 0)) // NL
        lp = params[len(EMBED_KEYS):] if embed else params
        ctx.meta = (n, L, H, R, heads, eps, embed, n_layers)
        ctx.idx, ctx.seg_off = idx, seg_off
        ctx.params = params
        out = torch.zeros(n * L, H, **f)
        if R == 0:
            ctx.layers, ctx.frame = [], None
            return out.view(n, L, H)
        if ROWS_SEEN is not None:
            ROWS_SEEN.append(R)
        A = arena_of(dev)
        ctx.frame = frame = A.push()
        xc = K.gather_rows(x.view(n * L, H), idx, out=A.take(R, H))
        saved = {}
        if embed:
            pos_w, type_w, g0, b0 = params[:4]
            pos = idx.long() % L                      # BertEmbeddings.position_ids: the row's place in its padded sequence (any mask)
            resid0 = torch.index_select(pos_w.detach(), 0, pos, out=A.take(R, H))          # position + token-type rows (token_type_ids = 0)
            resid0 += type_w.detach()[0]
            h = A.take(R, H)
            mean0, rstd0 = A.take(R), A.take(R)
            post, post_rng = _drop(p_hidden, training)
            _ln_fwd(xc, resid0, g0.detach(), b0.detach(), eps, None, post, h, mean0, rstd0)
            saved["embed"] = (xc, resid0, mean0, rstd0, post_rng, pos)
        else:
            h = xc
        layers = []
        I = lp[10].shape[0] if n_layers else 0
        for l in range(n_layers):
            Wq, bq, Wk, bk, Wv, bv, Wo, bo, g1, be1, W1, b1, W2, b2, g2, be2 = (t.detach() for t in lp[l * NL:(l + 1) * NL])
            # q, k, v as ONE product against the stacked [3H, H] weight (the three 768-wide launches cost 1.02 ms at 29.6 k rows, the
            # 2304-wide one 0.95; the data gradient 1.04 -> 0.88 ms: tools/bert_shapes_bench.py).  PluginStep lays the three tensors out
            # back to back in its flat parameter buffer: the stacked weight is then a VIEW; else a 7 MB copy into the arena
            Wqkv, bqkv = _stacked((Wq, Wk, Wv)), _stacked((bq, bk, bv))
            if Wqkv is None:
                Wqkv = torch.cat((Wq, Wk, Wv), 0, out=A.take(3 * H, H))
            if bqkv is None:
                bqkv = torch.cat((bq, bk, bv), 0, out=A.take(3 * H))
            qkv = A.take(R, 3 * H)
            _lin_fwd(h, Wqkv, bqkv, qkv, 3 * H, tag="qkv_fwd")
            ctxv = A.take(R, H)
            probs = A.take(R, heads, L)
            adrop, arng = _drop(p_attn, training)
            with _timed("mhsa_core_fwd", 4.0 * R * L * H / 2):            # ~L/2 live keys per row on average: nominal
                call("lego_mhsa_core_fwd", _ptr(qkv), 3 * H, _ptr(seg_off), n, None, H, heads, _ptr(ctxv), H, None, _ptr(probs), L,
                     adrop, R, 0, None, None, _stream())
            ao = A.take(R, H)
            _lin_fwd(ctxv, Wo, bo, ao, H, tag="attn_out_fwd")
            a = A.take(R, H)
            mean1, rstd1 = A.take(R), A.take(R)
            pre1, rng1 = _drop(p_hidden, training)
            _ln_fwd(ao, h, g1, be1, eps, pre1, None, a, mean1, rstd1)
            z = A.take(R, I)
            g = A.take(R, I)
            _ffn1_fwd(a, W1, b1, z, g)
            fo = A.take(R, H)
            _lin_fwd(g, W2, b2, fo, H, tag="ffn2_fwd")
            hn = A.take(R, H)
            mean2, rstd2 = A.take(R), A.take(R)
            pre2, rng2 = _drop(p_hidden, training)
            _ln_fwd(fo, a, g2, be2, eps, pre2, None, hn, mean2, rstd2)
            layers.append((h, qkv, probs, arng, ctxv, ao, a, mean1, rstd1, rng1, z, g, fo, mean2, rstd2, rng2, Wqkv))
            h = hn
        ctx.layers, ctx.saved = layers, saved
        out.index_copy_(0, idx.long(), h)
        if not any(ctx.needs_input_grad):            # no backward pass will come (evaluation): the workspace is free again
            ctx.layers, ctx.saved, ctx.frame = [], {}, None
            frame.release()
        return out.view(n, L, H)

    @staticmethod
    def backward(ctx, gout):
        n, L, H, R, heads, eps, embed, n_layers = ctx.meta
        params = ctx.params
        dev = gout.device
        f = dict(dtype=torch.float32, device=dev)
        n_fixed = 8
        need = ctx.needs_input_grad
        grads: List[Optional[torch.Tensor]] = [None] * len(params)
        dx_dense = torch.zeros(n * L, H, **f) if need[0] else None
        if R == 0:
            return (dx_dense.view(n, L, H) if need[0] else None,) + (None,) * (n_fixed - 1) + tuple(grads)
        frame = ctx.frame
        if frame is None or not frame.alive:
            raise RuntimeError("the BERT blocks' workspace frame is closed: a second backward pass through the same forward (retain_graph) is not supported")
        A = arena_of(dev)
        off = len(EMBED_KEYS) if embed else 0
        lp = params[off:]

        def want(i):                                 # gradient of params[i] wanted?
            return need[n_fixed + i]

        def direct(i):                               # ... and accumulated straight into params[i].grad (PluginStep's flat gradient buffer)?
            gr = params[i].grad
            return DIRECT_GRADS and gr is not None and gr.dtype == torch.float32 and gr.is_contiguous() and gr.shape == params[i].shape

        # every wanted gradient that is RETURNED is a view of ONE zero-filled buffer (one fill launch per backward instead of one per tensor;
        # the products and the LayerNorm column sums accumulate into it).  It leaves this pass (autograd may keep it as `.grad`): torch memory
        sizes = [(p.numel() + 3) // 4 * 4 if (need[n_fixed + i] and not direct(i)) else 0 for i, p in enumerate(params)]
        flat = torch.zeros(sum(sizes), **f) if sum(sizes) else None
        offs, o = [], 0
        for sz in sizes:
            offs.append(o)
            o += sz
        is_direct = [False] * len(params)

        def gbuf(i):
            if direct(i):
                is_direct[i] = True
                return params[i].grad
            grads[i] = flat[offs[i]:offs[i] + params[i].numel()].view(params[i].shape)
            return grads[i]
        I = lp[10].shape[0] if n_layers else 0
        dh = K.gather_rows(gout.detach().float().contiguous().view(n * L, H), ctx.idx, out=A.take(R, H))
        # backward temporaries, ONE set for all blocks (each is dead before the next block writes it)
        d_fo, d_a, d_ao = A.take(R, H), A.take(R, H), A.take(R, H)
        dg = A.take(R, I) if n_layers else None
        d_qkv = A.take(R, 3 * H) if n_layers else None
        for l in reversed(range(n_layers)):
            (h, qkv, probs, arng, ctxv, ao, a, mean1, rstd1, rng1, z, g, fo, mean2, rstd2, rng2, Wqkv) = ctx.layers[l]
            base = off + l * NL
            Wq, bq, Wk, bk, Wv, bv, Wo, bo, g1, be1, W1, b1, W2, b2, g2, be2 = (t.detach() for t in lp[l * NL:(l + 1) * NL])
            # ---- BertOutput: hn = LN(drop(fo) + a)
            _ln_bwd(dh, fo, a, g2, mean2, rstd2, _redrop(rng2), None, d_fo, d_a,
                    gbuf(base + 14) if want(base + 14) else None, gbuf(base + 15) if want(base + 15) else None,
                    gbuf(base + 13) if want(base + 13) else None)                         # ... and b2's gradient = colsum(d_fo), same pass
            if want(base + 12):
                _lin_bwd_weight(d_fo, H, 0, g, gbuf(base + 12), tag="ffn2_bwd_weight")
            _ffn2_bwd_data(d_fo, W2, z, dg)                                               # dz = (d_fo W2) . gelu'(z)
            if want(base + 10):
                _lin_bwd_weight(dg, I, 0, a, gbuf(base + 10), tag="ffn1_bwd_weight")
            if want(base + 11):
                _colsum(dg, I, 0, I, gbuf(base + 11))
            _lin_bwd_data(dg, I, 0, W1, d_a, True, tag="ffn1_bwd_data")                                        # d_a += dz W1
            # ---- BertSelfOutput: a = LN(drop(ao) + h)      (d_h takes dh's place: dh was consumed by the LayerNorm pass above)
            d_h = dh
            _ln_bwd(d_a, ao, h, g1, mean1, rstd1, _redrop(rng1), None, d_ao, d_h,
                    gbuf(base + 8) if want(base + 8) else None, gbuf(base + 9) if want(base + 9) else None,
                    gbuf(base + 7) if want(base + 7) else None)                           # bo's gradient = colsum(d_ao)
            if want(base + 6):
                _lin_bwd_weight(d_ao, H, 0, ctxv, gbuf(base + 6), tag="attn_out_bwd_weight")
            d_ctx = d_a                                                                   # reuse: d_a is consumed
            _lin_bwd_data(d_ao, H, 0, Wo, d_ctx, False, tag="attn_out_bwd_data")
            # ---- attention core: d(qkv) and, fused, the three bias gradients (column sums of d(qkv))
            any_b = want(base + 1) or want(base + 3) or want(base + 5)
            all_b = want(base + 1) and want(base + 3) and want(base + 5)
            bsum = _stacked(tuple(params[base + 1 + 2 * c].grad for c in range(3))) if (all_b and all(direct(base + 1 + 2 * c) for c in range(3))) else None
            b_direct = bsum is not None
            if any_b and not b_direct:
                bsum = torch.zeros(3 * H, **f)
            with _timed("mhsa_core_bwd", 8.0 * R * L * H / 2):
                call("lego_mhsa_core_bwd", _ptr(qkv), 3 * H, _ptr(ctx.seg_off), n, None, H, heads, _ptr(d_ctx), H, None, _ptr(probs), L,
                     _redrop(arng), R, _ptr(d_qkv), 3 * H, _ptr(bsum), 0, None, None, _stream())
            if want(base) and want(base + 2) and want(base + 4):          # the three weight gradients as one [3H, H] product
                gWqkv = _stacked(tuple(params[base + 2 * c].grad for c in range(3))) if all(direct(base + 2 * c) for c in range(3)) else None
                if gWqkv is not None:
                    for c in range(3):
                        is_direct[base + 2 * c] = True
                else:
                    gWqkv = torch.zeros(3 * H, H, **f)
                    for c in range(3):
                        grads[base + 2 * c] = gWqkv[c * H:(c + 1) * H]
                _lin_bwd_weight(d_qkv, 3 * H, 0, h, gWqkv, tag="qkv_bwd_weight")
            else:
                for c in range(3):
                    if want(base + 2 * c):
                        _lin_bwd_weight(d_qkv, 3 * H, c * H, h, gbuf(base + 2 * c))
            for c in range(3):
                if want(base + 2 * c + 1):
                    if b_direct:
                        is_direct[base + 2 * c + 1] = True
                    else:
                        grads[base + 2 * c + 1] = bsum[c * H:(c + 1) * H]
            _lin_bwd_data(d_qkv, 3 * H, 0, Wqkv, d_h, True, tag="qkv_bwd_data")               # d_h += d(qkv) Wqkv
            dh = d_h
        if embed:
            xc, resid0, mean0, rstd0, post_rng, pos = ctx.saved["embed"]
            pos_w, type_w, g0, b0 = params[:4]
            d_xc = A.take(R, H) if need[0] else None
            d_res = A.take(R, H) if (want(0) or want(1)) else None
            _ln_bwd(dh, xc, resid0, g0.detach(), mean0, rstd0, None, _redrop(post_rng), d_xc, d_res,
                    gbuf(2) if want(2) else None, gbuf(3) if want(3) else None)
            if want(0):
                gbuf(0).index_add_(0, pos, d_res)
            if want(1):
                gbuf(1)[0] += d_res.sum(0)
            dh = d_xc
        if need[0]:
            dx_dense.index_copy_(0, ctx.idx.long(), dh)
        ctx.layers, ctx.saved = [], {}
        frame.release()                              # (everything above is enqueued on this stream: the next pass may overwrite the workspace)
        return (dx_dense.view(n, L, H) if need[0] else None,) + (None,) * (n_fixed - 1) + tuple(None if is_direct[i] else g_ for i, g_ in enumerate(grads))


def encoder_forward(transformer, x, mask, embed: bool):
    """`transformer(inputs_embeds=x, attention_mask=mask).last_hidden_state` (embed=True) or
    `transformer.encoder(hidden_states=x, ...)` on the kept blocks (embed=False), live rows only; x [n, L, H], mask [n, L]"""
    cfg = transformer.config
    params = (embed_params(transformer) if embed else []) + layer_params(transformer)
    training = bool(transformer.training)
    return _Blocks.apply(x, mask.to(x.device), int(cfg.num_attention_heads), float(cfg.layer_norm_eps),
                         float(cfg.hidden_dropout_prob), float(cfg.attention_probs_dropout_prob), training, bool(embed), *params)
