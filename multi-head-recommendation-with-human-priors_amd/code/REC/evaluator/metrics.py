"""Ranking metrics over the [users, K+1] hit matrix (reference `code/REC/evaluator/metrics.py:17-41, 145-238`,
`base_metric.py:37-81`).  Host-side numpy: per-user curves @1..K are summed over the rank's users; the trainer
divides by the global user count after the all-reduce (reference trainer.py:1109-1123)."""
import numpy as np


class _TopkMetric:
    name = ''
    needs = ('rec.topk',)

    def __init__(self, config):
        self.topk = config['topk']
        self.num_prior_categories = config['eval_num_cats']
        self.eval_by_cat = config.get('eval_by_cat', True)
        self.eval_pred_len = config['eval_pred_len']
        self.outlier_user_metrics = config['outlier_user_metrics']
        self.int_to_category = config['int_to_category']

    def used_info(self, dataobject):
        rec = dataobject.get('rec.topk')
        rec = rec.numpy() if hasattr(rec, 'numpy') else np.asarray(rec)
        K = max(self.topk)
        return rec[:, :K].astype(bool), rec[:, K].astype(np.int64)

    def curve(self, hits, pos_len):
        raise NotImplementedError

    def _pack(self, curve, num_samples=None, prefix=None):
        s = curve.sum(axis=0)
        out = {}
        for k in self.topk:
            key = f'{self.name}@{k}' if prefix is None else f'{prefix}-{self.name}@{k}'
            out[key] = s[k - 1] if num_samples is None else (s[k - 1], num_samples)
        return out

    def calculate_metric(self, dataobject, pred_len=1):
        hits, pos_len = self.used_info(dataobject)
        out = self._pack(self.curve(hits, pos_len))
        if self.num_prior_categories > 1 and self.eval_by_cat and 'rec.tgt_tags' in dataobject:
            tgt = dataobject.get('rec.tgt_tags')
            tgt = tgt.numpy() if hasattr(tgt, 'numpy') else np.asarray(tgt)
            for c in range(self.num_prior_categories):
                m = tgt[:, c].astype(bool)
                out.update(self._pack(self.curve(hits[m], pos_len[m]), int(m.sum()), self.int_to_category[c]))
        if self.outlier_user_metrics is not None and pred_len == self.eval_pred_len - 1 and 'rec.outlier_users' in dataobject:
            o = dataobject.get('rec.outlier_users')
            o = (o.numpy() if hasattr(o, 'numpy') else np.asarray(o)).astype(bool)
            out.update(self._pack(self.curve(hits[o], pos_len[o]), int(o.sum()), f'outlier_{self.outlier_user_metrics}'))
        return out


class Recall(_TopkMetric):
    name = 'recall'

    def curve(self, hits, pos_len):
        return np.cumsum(hits, axis=1) / pos_len.reshape(-1, 1)


class NDCG(_TopkMetric):
    name = 'ndcg'

    def curve(self, hits, pos_len):
        K = hits.shape[1]
        disc = 1.0 / np.log2(np.arange(2, K + 2, dtype=np.float64))
        ideal = np.cumsum(disc)[np.minimum(np.arange(K)[None, :], np.minimum(pos_len, K)[:, None] - 1)]
        return np.cumsum(np.where(hits, disc[None, :], 0.0), axis=1) / ideal


class Entropy:
    """Category diversity of the recommended lists (shared across pred_len)."""
    name = 'entropy'
    needs = ('rec.topk',)

    def __init__(self, config):
        self.topk = config['topk']

    def calculate_metric(self, dataobject, pred_len=1):
        tags = dataobject.get('rec.rec_tags')
        tags = tags.numpy() if hasattr(tags, 'numpy') else np.asarray(tags)
        counts = np.cumsum(tags.astype(np.float64), axis=1)
        out = {}
        for k in self.topk:
            c = counts[:, k - 1, :]
            with np.errstate(divide='ignore', invalid='ignore'):
                p = c / c.sum(axis=1, keepdims=True)
                h = -np.sum(np.where(p > 0, p * np.log2(p), 0.0), axis=1)
            out[f'Entropy@{k}'] = h.sum(axis=0)
        return out


metrics_dict = {'recall': Recall, 'ndcg': NDCG, 'entropy': Entropy}
