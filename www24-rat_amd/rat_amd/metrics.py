"""AUC / logloss with the reference's definitions (fuxictr/metrics.py:22-41): roc_auc_score, and log_loss with
predictions clipped to [1e-7, 1-1e-7] (the ``eps=1e-7`` of the sklearn the reference pinned; newer sklearn
dropped that kwarg, so the clip is explicit here).  Host-side numpy: evaluation bookkeeping, not the hot path."""
import logging

import numpy as np


def auc_score(y_true, y_pred):
    y_true = np.asarray(y_true, dtype=np.float64).reshape(-1)
    y_pred = np.asarray(y_pred, dtype=np.float64).reshape(-1)
    order = np.argsort(y_pred, kind="mergesort")
    sorted_pred = y_pred[order]
    # average ranks over ties
    boundaries = np.concatenate([[True], sorted_pred[1:] != sorted_pred[:-1], [True]])
    starts = np.flatnonzero(boundaries[:-1])
    ends = np.flatnonzero(boundaries[1:])
    avg = 0.5 * (starts + ends) + 1.0
    group = np.cumsum(boundaries[:-1]) - 1
    ranks = np.empty_like(sorted_pred)
    ranks[order] = avg[group]
    pos = y_true == 1
    n_pos = float(pos.sum())
    n_neg = float(len(y_true) - n_pos)
    if n_pos == 0 or n_neg == 0:
        raise ValueError("Only one class present in y_true. ROC AUC score is not defined in that case.")
    return float((ranks[pos].sum() - n_pos * (n_pos + 1) / 2.0) / (n_pos * n_neg))


def log_loss(y_true, y_pred, eps=1e-7):
    y_true = np.asarray(y_true, dtype=np.float64).reshape(-1)
    p = np.clip(np.asarray(y_pred, dtype=np.float64).reshape(-1), eps, 1 - eps)
    return float(-(y_true * np.log(p) + (1 - y_true) * np.log(1 - p)).mean())


def evaluate_metrics(y_true, y_pred, metrics, **kwargs):
    result = dict()
    for metric in metrics:
        if metric in ("logloss", "binary_crossentropy"):
            result[metric] = log_loss(y_true, y_pred, eps=1e-7)
        elif metric == "AUC":
            result[metric] = auc_score(y_true, y_pred)
        else:
            raise NotImplementedError("metric=%s is outside the RAT_m2 hot path" % metric)
    logging.info("[Metrics] " + " - ".join("{}: {:.6f}".format(k, v) for k, v in result.items()))
    return result
