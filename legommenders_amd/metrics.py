"""Grouped ranking metrics of the evaluation path (mirror of the reference's utils/metrics.py:100-369:
GAUC = mean over `group_col` groups of roc_auc_score, NDCG@k = sklearn.ndcg_score, the reference's
non-standard MRR (mean over ALL positives of 1/rank, :144-160), fp32 mean over groups :367).
numpy on the host: scores arrive once per evaluation from the device (one D2H copy of [n_rows] floats)."""
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


def calculate(scores, labels, groups, names):
    """{name: value} for GAUC / MRR / NDCG@k, grouped by `groups` (user id, config/data/mind.yaml:24)."""
    scores = np.asarray(scores, dtype=np.float64)
    labels = np.asarray(labels)
    groups = np.asarray(groups)
    order = np.argsort(groups, kind="stable")
    g_sorted = groups[order]
    bounds = np.flatnonzero(np.r_[True, g_sorted[1:] != g_sorted[:-1], True])
    out = OrderedDict()
    for name in names:
        vals = []
        for a, b in zip(bounds[:-1], bounds[1:]):
            idx = order[a:b]
            l, s = labels[idx], scores[idx]
            if name == "GAUC":
                vals.append(_auc(l, s))
            elif name == "MRR":
                vals.append(_mrr(l, s))
            elif name.startswith("NDCG@"):
                vals.append(_ndcg(l, s, int(name.split("@")[1])))
            else:
                raise ValueError(f"metric {name} is outside the MI355X path (GAUC, MRR, NDCG@k are built)")
        out[name] = float(np.asarray(vals, dtype=np.float32).mean(dtype=np.float32))
    return out
