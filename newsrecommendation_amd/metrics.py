"""Ranking metrics of the eval loop (src/metrics.py:1-29, src/main.py:255-258): per-impression AUC / MRR / nDCG@k.
Own numpy implementation; AUC is the rank statistic (average ranks for ties) that sklearn's roc_auc_score computes."""
import numpy as np


def roc_auc_score(y_true, y_score):
    y_true = np.asarray(y_true)
    y_score = np.asarray(y_score, dtype=np.float64)
    n_pos = int(y_true.sum())
    n_neg = len(y_true) - n_pos
    if n_pos == 0 or n_neg == 0:
        raise ValueError("AUC needs both classes")
    order = np.argsort(y_score, kind="mergesort")
    s = y_score[order]
    # average 1-based rank of every tie group
    starts = np.r_[0, np.flatnonzero(s[1:] != s[:-1]) + 1]
    ends = np.r_[starts[1:], len(s)]
    ranks = np.empty(len(s), dtype=np.float64)
    for a, b in zip(starts, ends):
        ranks[order[a:b]] = 0.5 * (a + b - 1) + 1.0
    return (ranks[y_true == 1].sum() - n_pos * (n_pos + 1) / 2.0) / (n_pos * n_neg)


def dcg_score(y_true, y_score, k=10):
    order = np.argsort(y_score)[::-1]
    y_true = np.take(y_true, order[:k])
    return np.sum((2 ** y_true - 1) / np.log2(np.arange(len(y_true)) + 2))


def ndcg_score(y_true, y_score, k=10):
    return dcg_score(y_true, y_score, k) / dcg_score(y_true, y_true, k)


def mrr_score(y_true, y_score):
    order = np.argsort(y_score)[::-1]
    y_true = np.take(y_true, order)
    return np.sum(y_true / (np.arange(len(y_true)) + 1)) / np.sum(y_true)


def ctr_score(y_true, y_score, k=1):
    order = np.argsort(y_score)[::-1]
    return np.mean(np.take(y_true, order[:k]))
