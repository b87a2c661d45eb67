"""Evaluation path: representation caches + id-gather + dot + grouped metrics.

Reference: `Manager.setup(dev|test)` re-encodes ALL items (pages of cache_page_size=512, one un-batched
embedding look-up per item, loader/pager/fast_item_pager.py:100-104) and ALL users
(loader/cacher/user_cacher.py:63-97), then scores `[n_rows]` (user, item) pairs by two row gathers and a dot
(model/legommender.py:153-157,202-203,282) and hands them to MetricPool (base_lego.py:349-427).
Here: one ragged launch sequence per page of items / users, scores by gather + row-dot on device."""
from __future__ import annotations

import ctypes

import numpy as np
import torch

from . import metrics as M
from ._lib import call
from .engine import NamlEngine, NrmsEngine, _ptr, _stream


class Evaluator:
    def __init__(self, kind, params, data, item_page=512, user_page=512, heads=8, glove=True):
        self.kind, self.P, self.data = kind, params, data
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
        self.item_repr = torch.empty(n_items, D, dtype=torch.float32, device=dev)
        for s in range(0, n_items, self.item_page):
            ids = torch.arange(s, min(s + self.item_page, n_items), dtype=torch.int32, device=dev)
            self.item_repr[s:s + ids.numel()] = self.item_eng.item_vectors(ids)
        self.user_repr = torch.empty(n_users, D, dtype=torch.float32, device=dev)
        for s in range(0, n_users, self.user_page):
            e = min(s + self.user_page, n_users)
            self.user_repr[s:e] = self.user_eng.user_vectors(self.item_repr, d.user_hist[s:e], d.user_hist_len[s:e])
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
        self.build_caches()
        s = self.scores(torch.as_tensor(users), torch.as_tensor(items))
        g = np.asarray(users if groups is None else groups)
        return M.calculate_device(s, np.asarray(labels), g, list(metrics)), s.cpu().numpy()
