"""CPU oracle of the multi-head decode (test infrastructure only).

Restates reference `code/REC/evaluator/collector.py:153-325` (per-head top-k,
cross-head merge, first-occurrence dedup, hit matrix) and the history / pad
suppression of `code/REC/trainer/trainer.py:724-726`, in numpy.

Tie-breaking, which the reference leaves to torch.topk / an unstable torch.sort:
  * per-head top-k:  value descending, then item index ascending;
  * cross-head merge: value descending, then (head, rank-in-head) ascending.
For inputs without exact ties among finite candidates both orders coincide with
the reference's; -inf candidates (heads with fewer than K admissible items,
SURVEY.md H2) are ordered by this rule, where the reference's order is arbitrary.
"""
import numpy as np


def suppress(scores, hist_u=None, hist_i=None):
    """scores[:, :, 0] = -inf; scores[hist_u, :, hist_i] = -inf (in place). trainer.py:724-726"""
    scores[:, :, 0] = -np.inf
    if hist_u is not None and len(hist_u):
        scores[hist_u, :, hist_i] = -np.inf
    return scores


def topk_desc(row, k):
    """Indices/values of the k largest entries: value desc, index asc on ties."""
    n = row.shape[0]
    if k < n:
        kth = np.partition(row, n - k)[n - k]                          # k-th largest value
        cand = np.nonzero(row >= kth)[0]                               # all ties at the boundary included
    else:
        cand = np.arange(n)
    order = np.lexsort((cand, -row[cand].astype(np.float64)))          # primary: -value, secondary: index
    idx = cand[order[:k]]
    return idx.astype(np.int64), row[idx]


def per_head_topk(scores, k):
    """scores [B,H,N] float32 -> (values [B,H,k], indices [B,H,k]). collector.py:245"""
    B, H, _ = scores.shape
    vals = np.empty((B, H, k), np.float32)
    idxs = np.empty((B, H, k), np.int64)
    for b in range(B):
        for h in range(H):
            i, v = topk_desc(scores[b, h], k)
            idxs[b, h], vals[b, h] = i, v
    return vals, idxs


def merge_dedup(vals, idxs, k):
    """Cross-head merge. collector.py:249-275.

    vals/idxs [B,H,K'] -> merged (items [B,k] i64, values [B,k] f32, head_source [B,k] i64).
    Flatten (head-major), sort by value descending, keep the first occurrence of
    every item, take the first k.  Raises if a row has fewer than k unique items
    (the reference's torch.stack over ragged rows fails in the same case).
    """
    B, H, Kp = vals.shape
    items = np.empty((B, k), np.int64)
    values = np.empty((B, k), np.float32)
    source = np.empty((B, k), np.int64)
    flat_src = np.repeat(np.arange(H), Kp)
    for b in range(B):
        fv = vals[b].reshape(-1)
        fi = idxs[b].reshape(-1)
        order = np.lexsort((np.arange(H * Kp), -fv.astype(np.float64)))
        seen = set()
        out = []
        for j in order:
            it = int(fi[j])
            if it in seen:
                continue
            seen.add(it)
            out.append(j)
            if len(out) == k:
                break
        if len(out) < k:
            raise ValueError(f"row {b}: only {len(out)} unique candidates < k={k}")
        out = np.asarray(out)
        items[b], values[b], source[b] = fi[out], fv[out], flat_src[out]
    return items, values, source


def average_heads_topk(scores, k):
    """split_mode='average'. collector.py:227-230: mean over finite heads, then top-k."""
    finite = np.isfinite(scores)
    s = np.where(finite, scores, 0).sum(1) / (finite.sum(1) + 1e-8)
    s = s.astype(np.float32)
    out = np.stack([topk_desc(s[b], k)[0] for b in range(s.shape[0])])
    return out


def decode_topk(scores, k, split_mode="combine"):
    """collector.py:203-285 -> merged item ids [B,k]."""
    if scores.shape[1] == 1:
        return np.stack([topk_desc(scores[b, 0], k)[0] for b in range(scores.shape[0])])
    if split_mode == "average":
        return average_heads_topk(scores, k)
    if split_mode == "combine":
        v, i = per_head_topk(scores, k)
        return merge_dedup(v, i, k)[0]
    raise ValueError(f"Unknown split_mode: {split_mode}")


def hit_matrices(topk_idx, positive_i, pred_len_list):
    """collector.py:300-316 -> {pred_idx: uint8/int [B, K+1]}.

    Column K holds pos_len: the number of distinct values among the first pred_idx+1
    entries of the row of positives *sorted ascending over the whole row* (the
    reference sorts all eval_pred_len targets before the cumulative count).
    The hit mask is cumulative over pred_len_list with positives [0 : pred_idx+1].
    """
    B, K = topk_idx.shape
    srt = np.sort(positive_i, axis=1)
    first = np.ones_like(srt, dtype=bool)
    first[:, 1:] = srt[:, 1:] != srt[:, :-1]
    pos_len_full = np.cumsum(first, axis=1).astype(np.int32)
    out = {}
    hit = np.zeros((B, K), bool)
    for p in pred_len_list:
        sl = positive_i[:, 0:p + 1]
        hit |= (topk_idx[:, :, None] == sl[:, None, :]).any(-1)
        out[p] = np.concatenate([hit.astype(np.int32), pos_len_full[:, p:p + 1]], axis=1)
    return out
