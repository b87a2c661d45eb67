"""Metrics of the evaluation path (mirror of the reference's MetricPool, utils/metrics.py:60-369).

Group-wise metrics (GAUC, MRR, MRR0, LRAP, NDCG@k, HitRatio@k, Recall@k: mean over `group_col` groups, fp32 mean :367)
run on the device: `calculate_device` sorts the rows by group once and `lego_grouped_metrics` evaluates every group
and every requested metric in one launch; only the [n_metrics, n_groups] table comes back.  Point-wise metrics
(AUC, LogLoss, F1@t over all rows, :66-96,162-181) are three numpy lines on the host copy of the scores.

`calculate` is the host form of the same definitions for callers that hold host arrays (the reference's MetricPool
is host code); the evaluators of this package use `calculate_device`."""
from __future__ import annotations

from collections import OrderedDict

import numpy as np

MINIMIZE = {"LogLoss"}


def is_minimize(name: str) -> bool:
    return name.split("@")[0] in MINIMIZE


def _auc(labels, scores):
    order = np.argsort(scores, kind="mergesort")
    s = scores[order]
    ranks = np.empty(len(s), dtype=np.float64)
    i = 0
    while i < len(s):                    # average ranks over ties (Mann-Whitney U == sklearn's trapezoid AUC)
        j = i
        while j + 1 < len(s) and s[j + 1] == s[i]:
            j += 1
        ranks[i:j + 1] = 0.5 * (i + j) + 1.0
        i = j + 1
    r = np.empty_like(ranks)
    r[order] = ranks
    pos = labels == 1
    n_pos, n_neg = int(pos.sum()), int((~pos).sum())
    if n_pos == 0 or n_neg == 0:
        raise ValueError("Only one class present in y_true. ROC AUC score is not defined in that case.")
    return (r[pos].sum() - n_pos * (n_pos + 1) / 2.0) / (n_pos * n_neg)


def _dcg(rel, scores, k):
    disc = 1.0 / np.log2(np.arange(len(rel)) + 2.0)
    disc[k:] = 0.0
    _, inv, counts = np.unique(-scores, return_inverse=True, return_counts=True)
    ranked = np.zeros(len(counts))
    np.add.at(ranked, inv, rel)
    ranked /= counts
    groups = np.cumsum(counts) - 1
    cs = np.cumsum(disc)
    dsum = np.empty(len(counts))
    dsum[0] = cs[groups[0]]
    dsum[1:] = np.diff(cs[groups])
    return float((ranked * dsum).sum())


def _ndcg(labels, scores, k):
    rel = labels.astype(np.float64)
    ideal = _dcg(rel, rel, k)
    return _dcg(rel, scores.astype(np.float64), k) / ideal if ideal > 0 else 0.0


def _mrr(labels, scores):
    order = np.argsort(-scores, kind="stable")
    y = labels[order]
    return float((y / (np.arange(len(y)) + 1.0)).sum() / y.sum())


def _ranks_desc(scores):
    """0-based position of every row after a stable descending sort (Python `sorted(..., reverse=True)`)"""
    order = np.argsort(-scores, kind="stable")
    r = np.empty(len(scores), dtype=np.int64)
    r[order] = np.arange(len(scores))
    return r


def _mrr0(labels, scores):
    r = _ranks_desc(scores)[labels == 1]
    return 1.0 / (r.min() + 1.0) if r.size else 0.0


def _hit(labels, scores, k):
    return float(((_ranks_desc(scores) < k) & (labels == 1)).any())


def _recall(labels, scores, k):
    return float(((_ranks_desc(scores) < k) & (labels == 1)).sum() / (labels == 1).sum())


def _lrap(labels, scores):
    pos = labels == 1
    if not pos.any() or pos.all():
        return 1.0
    sp = scores[pos]
    return float(np.mean([(scores[pos] >= v).sum() / (scores >= v).sum() for v in sp]))


def _pointwise(name, scores, labels):
    """AUC / LogLoss / F1@t over all rows (utils/metrics.py:66-96,162-181)"""
    base, _, arg = name.partition("@")
    if base.upper() == "AUC":
        return float(_auc(labels, scores))
    if base.upper() == "LOGLOSS":                       # sklearn.log_loss: probabilities clipped to [eps, 1 - eps] of the dtype
        if scores.min() < 0.0 or scores.max() > 1.0:
            raise ValueError("y_prob contains values outside [0, 1] (LogLoss needs probabilities)")
        eps = np.finfo(np.float64).eps
        p = np.clip(scores, eps, 1.0 - eps)
        return float(-np.mean(np.where(labels == 1, np.log(p), np.log1p(-p))))
    if base.upper() == "F1":
        pred = scores >= (float(arg) if arg else 0.5)
        tp = float((pred & (labels == 1)).sum())
        denom = float(pred.sum() + (labels == 1).sum())
        return 2.0 * tp / denom if denom > 0 else 0.0
    raise ValueError(f"Metric {base} not found")


POINTWISE = {"AUC", "LOGLOSS", "F1"}
GROUPED_ROWS = {"GAUC": 0, "MRR": 1, "MRR0": 2, "LRAP": 3}
GROUPED_K = {"NDCG": 0, "HITRATIO": 1, "RECALL": 2}
MAX_K = 8                                               # LEGO_METRIC_MAX_K


def calculate_device(scores, labels, groups, names):
    """{name: value} with the grouped metrics evaluated by `lego_grouped_metrics`; `scores` is a device fp32 tensor
    [n_rows] (it stays on the device for the grouped metrics), labels / groups are host arrays."""
    import torch
    from ._lib import call
    from .engine import _ptr, _stream
    names = list(names)
    labels = np.asarray(labels)
    groups = np.asarray(groups)
    dev = scores.device
    out = OrderedDict()
    grouped = [n for n in names if n.partition("@")[0].upper() not in POINTWISE]
    table, ks = None, []
    if grouped:
        for n in grouped:
            base, _, arg = n.partition("@")
            if base.upper() in GROUPED_K:
                if int(arg) not in ks:
                    ks.append(int(arg))
            elif base.upper() not in GROUPED_ROWS:
                raise ValueError(f"Metric {base} not found")
        if len(ks) > MAX_K:
            raise ValueError(f"at most {MAX_K} distinct cut-offs per evaluation")
        order = np.argsort(groups, kind="stable")
        g_sorted = groups[order]
        off = np.flatnonzero(np.r_[True, g_sorted[1:] != g_sorted[:-1], True]).astype(np.int32)
        G = off.size - 1
        s_sorted = scores.to(torch.float32)[torch.as_tensor(order, device=dev)].contiguous()
        l_dev = torch.as_tensor(labels[order].astype(np.int32), device=dev)
        off_dev = torch.as_tensor(off, device=dev)
        table = torch.empty(4 + 3 * len(ks), G, dtype=torch.float32, device=dev)
        ks_host = np.asarray(ks, dtype=np.int32)
        call("lego_grouped_metrics", _ptr(s_sorted), _ptr(l_dev), _ptr(off_dev), G,
             ks_host.ctypes.data if ks else None, len(ks), _ptr(table), _stream())
        table = table.cpu().numpy()
    host_scores = None
    for n in names:
        base, _, arg = n.partition("@")
        if base.upper() in POINTWISE:
            if host_scores is None:
                host_scores = scores.detach().cpu().numpy().astype(np.float64)
            out[n] = _pointwise(n, host_scores, labels)
            continue
        row = GROUPED_ROWS[base.upper()] if base.upper() in GROUPED_ROWS else 4 + 3 * ks.index(int(arg)) + GROUPED_K[base.upper()]
        vals = table[row]
        if np.isnan(vals).any():                        # sklearn raises / the reference divides by zero for such a group
            raise ValueError(f"{n}: a group holds one class only")
        out[n] = float(vals.mean(dtype=np.float32))
    return out


def calculate(scores, labels, groups, names):
    """{name: value} on host arrays, grouped by `groups` (user id, config/data/mind.yaml:24)."""
    scores = np.asarray(scores, dtype=np.float64)
    labels = np.asarray(labels)
    groups = np.asarray(groups)
    order = np.argsort(groups, kind="stable")
    g_sorted = groups[order]
    bounds = np.flatnonzero(np.r_[True, g_sorted[1:] != g_sorted[:-1], True])
    out = OrderedDict()
    for name in names:
        if name.partition("@")[0].upper() in POINTWISE:
            out[name] = _pointwise(name, scores, labels)
            continue
        vals = []
        for a, b in zip(bounds[:-1], bounds[1:]):
            idx = order[a:b]
            l, s = labels[idx], scores[idx]
            if name == "GAUC":
                vals.append(_auc(l, s))
            elif name == "MRR":
                vals.append(_mrr(l, s))
            elif name == "MRR0":
                vals.append(_mrr0(l, s))
            elif name == "LRAP":
                vals.append(_lrap(l, s))
            elif name.upper().startswith("NDCG@"):
                vals.append(_ndcg(l, s, int(name.split("@")[1])))
            elif name.upper().startswith("HITRATIO@"):
                vals.append(_hit(l, s, int(name.split("@")[1])))
            elif name.upper().startswith("RECALL@"):
                vals.append(_recall(l, s, int(name.split("@")[1])))
            else:
                raise ValueError(f"Metric {name} not found")
        out[name] = float(np.asarray(vals, dtype=np.float32).mean(dtype=np.float32))
    return out
