"""The reference's dense operator shapes on `torch.ops.lego_hip.*` (legommenders_amd/ops.py: the HIP kernels registered
with the PyTorch dispatcher, autograd and fake implementations included).

This is the layer the plug-in classes (`model/operators/*`, `model/predictors/*`, `loader/embedding_hub`) call, so
third-party operators written against the reference's interface keep working while every arithmetic step runs in
liblego_hip.so.  The functions here only reshape to the ops' 2-D / 3-D layouts and hand each dropout call its Philox
stream `(p, seed, site)`.  The training fast path (`engine.py`) bypasses autograd entirely.  No CPU fallback: the ops have
ROCm implementations only, and CPU tensors raise `LegoHipError` here before they reach the dispatcher.
"""
from __future__ import annotations

import itertools

import torch

from . import ops as _ops  # noqa: F401  (registers torch.ops.lego_hip)
from ._lib import LegoHipError

_site = itertools.count(1)
SEED = 2023
L = torch.ops.lego_hip


def seed_streams(seed: int):
    """seed of the Philox dropout streams of this process (data parallel: train_step.rank_seed(seed, rank))"""
    global SEED
    SEED = int(seed)


def _rng(p, training):
    """(p, seed, site): a fresh Philox stream per call; p = 0 in eval"""
    if not training or p <= 0:
        return 0.0, 0, 0
    return float(p), SEED, 1000 + next(_site)


def _need_gpu(t, what):
    if not t.is_cuda:
        raise LegoHipError(f"{what}: CPU tensors are not supported -- the MI355X path has no CPU fallback "
                           "(use the reference for --cuda -1)")


def linear(x, W, b=None, act=0):
    _need_gpu(x, "linear")
    if act != 0:
        raise LegoHipError("functional.linear implements nn.Linear only; tanh is fused inside additive_attention")
    shape = x.shape
    y = L.linear(x.reshape(-1, shape[-1]), W, b)
    return y.view(*shape[:-1], W.shape[0])


def glove_project(ids, table, W, b, p=0.0, training=False):
    """Transformation.forward + SimpleInputer masking (loader/embedding_hub.py:95-96, simple_inputer.py:55-63)"""
    _need_gpu(ids, "embedding")
    H, _ = L.glove_project(ids.reshape(-1), table, W, b, *_rng(p, training))
    return H.view(*ids.shape, W.shape[0])


def embedding(ids, table):
    """trainable nn.Embedding look-up with the inputer's pad handling (id -1 -> zero row), dense gradient"""
    _need_gpu(ids, "embedding")
    return L.gather_rows(table, ids.reshape(-1)).view(*ids.shape, table.shape[1])


def conv3_relu_mask(h, mask, w, b, p=0.0, training=False):
    """CNNOperator title branch: Conv1d(k=3,'same') -> ReLU -> *mask -> Dropout (cnn_operator.py:54-57)"""
    _need_gpu(h, "conv3")
    if w.shape[2] != 3:
        raise LegoHipError("the HIP conv kernel implements kernel_size=3 (config/model/naml.yaml:14)")
    return L.conv3_relu_mask(h, mask, w, b, *_rng(p, training))


def additive_attention(x, mask, W1, b1, w2):
    _need_gpu(x, "additive_attention")
    return L.additive_pool(x, mask, W1, b1, w2)[0]


def multi_head_self_attention(x, mask, in_w, in_b, out_w, out_b, heads, p=0.0, training=False):
    _need_gpu(x, "multi_head_attention")
    return L.mhsa(x, mask, in_w, in_b, out_w, out_b, int(heads), *_rng(p, training))[0]


def rowdot(u, it):
    _need_gpu(u, "dot predictor")
    return L.rowdot(u, it)


def dot_ce(user, items):
    """fused DotPredictor + CrossEntropy(label 0) on [B,D] users x [B,C,D] candidates -> (loss, scores)"""
    _need_gpu(user, "dot_ce")
    return L.dot_ce(user, items)
