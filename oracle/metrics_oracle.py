"""CPU oracle of the ranking metrics (test infrastructure only).

Restates reference `code/REC/evaluator/metrics.py` (Recall 145-181, NDCG 184-238,
Entropy 17-41) and `base_metric.py:51-81`: per-user curves @1..K summed over users;
the trainer later divides by the global user count (`trainer.py:1109-1123`).
"""
import numpy as np


def split_topk(rec_mat, K):
    """base_metric.py:51-57: [users, K+1] -> (hits bool [users,K], pos_len [users])"""
    return rec_mat[:, :K].astype(bool), rec_mat[:, K].astype(np.int64)


def recall_curve(hits, pos_len):
    return np.cumsum(hits, axis=1) / pos_len.reshape(-1, 1)


def ndcg_curve(hits, pos_len):
    users, K = hits.shape
    disc = 1.0 / np.log2(np.arange(1, K + 1, dtype=np.float64) + 1)
    ideal_len = np.minimum(pos_len, K)
    idcg_all = np.cumsum(disc)
    # idcg[u, j] = sum of the first min(j+1, ideal_len[u]) discounts
    cols = np.minimum(np.arange(K)[None, :], ideal_len[:, None] - 1)
    idcg = idcg_all[cols]
    dcg = np.cumsum(np.where(hits, disc[None, :], 0.0), axis=1)
    return dcg / idcg


def summed_at(curve, topk):
    s = curve.sum(axis=0)
    return {k: s[k - 1] for k in topk}


def recall_ndcg(rec_mat, topk):
    """-> {'recall@k': sum over users, 'ndcg@k': ...}"""
    K = max(topk)
    hits, pos_len = split_topk(rec_mat, K)
    out = {}
    for name, fn in (("recall", recall_curve), ("ndcg", ndcg_curve)):
        for k, v in summed_at(fn(hits, pos_len), topk).items():
            out[f"{name}@{k}"] = v
    return out


def entropy(rec_tags, topk):
    """metrics.py:29-41.  rec_tags [users, K, C] bool -> {'Entropy@k': sum over users}"""
    counts = np.cumsum(rec_tags * 1.0, axis=1)
    out = {}
    for k in topk:
        c = counts[:, k - 1, :]
        p = c / c.sum(axis=1, keepdims=True)
        with np.errstate(divide="ignore", invalid="ignore"):
            h = -np.sum(np.where(p > 0, p * np.log2(p), 0.0), axis=1)
        out[f"Entropy@{k}"] = h.sum(axis=0)
    return out
