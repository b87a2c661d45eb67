"""Evaluation path: representation caches + id-gather + dot + grouped metrics.

Reference: `Manager.setup(dev|test)` re-encodes ALL items (pages of cache_page_size=512, one un-batched
embedding look-up per item, loader/pager/fast_item_pager.py:100-104) and ALL users
(loader/cacher/user_cacher.py:63-97), then scores `[n_rows]` (user, item) pairs by two row gathers and a dot
(model/legommender.py:153-157,202-203,282) and hands them to MetricPool (base_lego.py:349-427).
Here: one ragged launch sequence per page of items / users, scores by gather + row-dot on device.

Multi-GPU: every rank encodes a contiguous shard of the items, one `all_gather` (RCCL) makes the item
cache whole on every rank, then the same for the users (whose encoder reads `item_repr[history]`); scoring and the
metrics run on rank 0."""
from __future__ import annotations

import ctypes

import numpy as np
import torch

from . import metrics as M
from ._lib import call
from .engine import NamlEngine, NrmsEngine, _ptr, _stream


def shard_bounds(n: int, rank: int, world: int):
    """contiguous shard [lo, hi) of `n` rows for `rank`, equal capacity `per` = ceil(n / world) (the tail shard is short)"""
    per = -(-n // world)
    lo = min(rank * per, n)
    return lo, min(lo + per, n), per


def gather_shards(local: torch.Tensor, n: int, process_group, world: int) -> torch.Tensor:
    """local: [per, D] with this rank's rows first -> the whole [n, D] on every rank (one all_gather)"""
    if process_group is None:
        return local[:n]
    full = torch.empty(world * local.shape[0], local.shape[1], dtype=local.dtype, device=local.device)
    torch.distributed.all_gather_into_tensor(full, local.contiguous(), group=process_group)
    return full[:n]


class Evaluator:
    def __init__(self, kind, params, data, item_page=512, user_page=512, heads=8, glove=True, process_group=None,
                 rank=0, world_size=1):
        self.kind, self.P, self.data = kind, params, data
        self.pg, self.rank, self.world = process_group, rank, world_size
        tb = data.tables
        mk = (lambda **kw: NamlEngine(params, tb, p_proj=0.0, p_conv=0.0, **kw)) if kind == "naml" else \
             (lambda **kw: NrmsEngine(params, tb, heads=heads, glove=glove, p_proj=0.0, p_att=0.0, **kw))
        self.item_eng = mk(B=1, C=item_page, S=0)
        self.user_eng = mk(B=user_page, C=0, S=data.S, token_rows=False)
        self.item_page, self.user_page = item_page, user_page
        self.item_repr = None
        self.user_repr = None

    @torch.no_grad()
    def build_caches(self):
        d, dev = self.data, self.data.tables.title_tok.device
        n_items, n_users = d.n_items, d.user_hist.shape[0]
        D = self.item_eng.D
        lo, hi, per = shard_bounds(n_items, self.rank, self.world)
        local = torch.zeros(per, D, dtype=torch.float32, device=dev)
        for s in range(lo, hi, self.item_page):
            ids = torch.arange(s, min(s + self.item_page, hi), dtype=torch.int32, device=dev)
            local[s - lo:s - lo + ids.numel()] = self.item_eng.item_vectors(ids)
        self.item_repr = gather_shards(local, n_items, self.pg, self.world)
        lo, hi, per = shard_bounds(n_users, self.rank, self.world)
        local = torch.zeros(per, D, dtype=torch.float32, device=dev)
        for s in range(lo, hi, self.user_page):
            e = min(s + self.user_page, hi)
            local[s - lo:e - lo] = self.user_eng.user_vectors(self.item_repr, d.user_hist[s:e], d.user_hist_len[s:e])
        self.user_repr = gather_shards(local, n_users, self.pg, self.world)
        return self.item_repr, self.user_repr

    @torch.no_grad()
    def scores(self, users: torch.Tensor, items: torch.Tensor) -> torch.Tensor:
        """score[r] = <user_repr[users[r]], item_repr[items[r]]> for n_rows evaluation rows"""
        dev = self.item_repr.device
        u = users.to(dev, torch.int32).contiguous()
        it = items.to(dev, torch.int32).contiguous()
        n, D = u.numel(), self.item_repr.shape[1]
        gu = torch.empty(n, D, dtype=torch.float32, device=dev)
        gi = torch.empty(n, D, dtype=torch.float32, device=dev)
        out = torch.empty(n, dtype=torch.float32, device=dev)
        st = _stream()
        call("lego_gather_rows", _ptr(self.user_repr), D, D, _ptr(u), n, None, _ptr(gu), D, 0, st)
        call("lego_gather_rows", _ptr(self.item_repr), D, D, _ptr(it), n, None, _ptr(gi), D, 0, st)
        call("lego_rowdot_fwd", _ptr(gu), D, _ptr(gi), D, n, D, _ptr(out), st)
        return out

    def evaluate(self, users, items, labels, groups=None, metrics=("GAUC", "MRR", "NDCG@1", "NDCG@5", "NDCG@10")):
        """every rank calls this (the cache build holds the collectives); ranks other than 0 return ({}, None)"""
        self.build_caches()
        if self.rank != 0:
            return {}, None
        s = self.scores(torch.as_tensor(users), torch.as_tensor(items))
        g = np.asarray(users if groups is None else groups)
        return M.calculate_device(s, np.asarray(labels), g, list(metrics)), s.cpu().numpy()
