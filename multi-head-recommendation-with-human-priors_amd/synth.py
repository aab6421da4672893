"""Synthetic Pixel8M-shaped data with the reference's tensor contract (no dataset exists
offline).  Generated on the device so the generator never throttles the step being measured.

Contract (reference `code/REC/data/dataset/trainset.py:155-177`, `collate_fn.py:59-90`):
  train batch = (items [B,L+P] i64, neg_items [B,C+1 or 1,n_neg] i64, mask [B,L+P] i64, tags [B,L+P,C] i64 or [B,0])
  eval batch  = (user_ids, item_seq [B,L] front zero-padded, item_target [B,E], (history_u, history_i), positive_u,
                 time_seq, target_tags [B,E,C], outlier_users)
ids ~ Zipf(1.05) through a fixed permutation; valid length ~ U{ceil(L/4)..L}; pad slots of a train window hold random
real ids with mask 0 (`pad_random_sample`); each item joins each category w.p. 0.375, at least one forced;
negatives uniform from the category pool / the global pool; n_neg = ceil(num_negatives / world / B).
"""
import math
import os

import numpy as np
import torch


class SyntheticData:
    def __init__(self, config, item_num, device, seed=2020, rank=0, world=1):
        self.config = config
        self.item_num = N = item_num
        self.device = device
        self.L = config['MAX_ITEM_LIST_LENGTH']
        self.P = config['pred_len']
        self.E = config['eval_pred_len']
        self.C = C = config['num_prior_head'] if config['loss'] == 'prior' else 1
        self.by_event = (config['category_by'] or 'item') == 'event'     # categories = interaction event types (MerRec)
        self.by_cat = bool(config['neg_sample_by_cat']) and config['loss'] == 'prior' and not self.by_event
        self.world = world
        g = np.random.default_rng(seed)                       # item-level structure is the same on every rank
        tags = g.random((N, C)) < 0.375
        tags[np.arange(N), g.integers(0, C, N)] = True
        if (config['category_by'] or 'item') == 'event':      # every item admissible in every head (batchset.py:36-38)
            tags[:] = True
        tags[0] = False
        self.item_tags = torch.from_numpy(tags).to(device)                              # [N,C] bool
        self.category_to_int = {f"cat{c}": c for c in range(C)}
        self.int_to_category = {c: f"cat{c}" for c in range(C)}
        self.category_counts = {f"cat{c}": int(tags[:, c].sum()) for c in range(C)}
        w = 1.0 / np.arange(1, N, dtype=np.float64) ** 1.05
        self.cdf = torch.from_numpy(np.cumsum(w / w.sum())).to(device)
        self.perm = torch.from_numpy(g.permutation(np.arange(1, N))).to(device)         # rank -> item id
        self.cat_items = [torch.nonzero(self.item_tags[:, c]).flatten() for c in range(C)]
        self.gen = torch.Generator(device=device)
        self.gen.manual_seed(seed + 1000003 * (rank + 1))

    def _zipf(self, shape):
        u = torch.rand(shape, generator=self.gen, device=self.device, dtype=torch.float64)
        r = torch.searchsorted(self.cdf, u).clamp_(max=self.item_num - 2)
        return self.perm[r]

    def _uniform_items(self, shape):
        return torch.randint(1, self.item_num, shape, generator=self.gen, device=self.device)

    def n_neg(self, batch_size):
        nn_ = self.config['num_negatives']
        return math.ceil(nn_ / self.world / batch_size) if nn_ else self.L

    def train_batch(self, B):
        L, P, C, dev = self.L, self.P, self.C, self.device
        W = L + P
        items = self._zipf((B, W))
        ctx_len = torch.randint(math.ceil(L / 4), L + 1, (B,), generator=self.gen, device=dev)
        pos = torch.arange(W, device=dev)[None, :]
        mask = (pos >= (L - ctx_len)[:, None]).long()                                    # front padding, full pred window
        items = torch.where(mask.bool(), items, self._uniform_items((B, W)))             # pad slots: random ids, mask 0
        n_neg = self.n_neg(B)
        if self.by_cat:
            pools = []
            for c in range(C):
                pool = self.cat_items[c]
                pools.append(pool[torch.randint(0, pool.numel(), (B, n_neg), generator=self.gen, device=dev)])
            pools.append(self._uniform_items((B, n_neg)))
            neg = torch.stack(pools, dim=1)
        else:
            neg = self._uniform_items((B, 1, n_neg))
        if self.config['loss'] == 'prior' and self.by_event:   # one event type per interaction (trainset.py:137-143)
            ev = torch.randint(0, C, (B, W), generator=self.gen, device=dev)
            tags = torch.nn.functional.one_hot(ev, C).long() * mask[..., None]
        elif self.config['loss'] == 'prior':
            tags = self.item_tags[items].long()
        else:
            tags = torch.zeros(B, 0, dtype=torch.long, device=dev)
        if getattr(self, "rows_hint", False):
            # what a host-side loader knows when it builds the batch (the reference's collate runs on the CPU: trainset.py:111-137):
            # how many valid positions the windows hold - as a static capacity, rounded up to ROWS_BUCKET, for the packed encoder
            mask._mhr_rows_cap = rows_capacity(int(ctx_len.sum()), n_rows=B * L)
        return items, neg, mask, tags

    def eval_batch(self, B, hist_extra=40):
        L, E, dev = self.L, self.E, self.device
        seq = self._zipf((B, L))
        ctx_len = torch.randint(math.ceil(L / 4), L + 1, (B,), generator=self.gen, device=dev)
        pos = torch.arange(L, device=dev)[None, :]
        item_seq = torch.where(pos >= (L - ctx_len)[:, None], seq, torch.zeros_like(seq))
        if getattr(self, "rows_hint", False):            # (eval steps are host-issued: no graph per capacity, fine buckets)
            item_seq._mhr_rows_cap = rows_capacity(int(ctx_len.sum()), n_rows=B * L, bucket=256)
        target = self._zipf((B, E))
        # full history = the visible sequence plus older interactions that fell out of the window
        older = self._zipf((B, hist_extra))
        keep = item_seq != 0
        hu = torch.cat([torch.nonzero(keep)[:, 0], torch.arange(B, device=dev).repeat_interleave(hist_extra)])
        hi = torch.cat([item_seq[keep], older.reshape(-1)])
        if self.by_event:
            target_tags = torch.nn.functional.one_hot(torch.randint(0, self.C, (B, E), generator=self.gen, device=dev), self.C).long()
        else:
            target_tags = self.item_tags[target].long()
        positive_u = torch.arange(B, device=dev)[:, None].repeat(1, E)
        users = torch.arange(B, device=dev)
        outlier = torch.zeros(B, dtype=torch.bool, device=dev)
        return users, item_seq, target, (hu, hi), positive_u, [], target_tags, outlier


def rows_capacity(n_valid, n_rows=None, bucket=None):
    """Static row capacity for a batch with n_valid valid window positions out of n_rows: the next multiple of `bucket`.  Every
    capacity is its own captured step graph, and the step's time is not monotone in the row count (measured at cfg1: capacities
    of 15 360 / 17 408 rows lose to 16 384 / 18 432 - library GEMM heuristics, stream counts), so the default bucket is coarse
    and a power of two: the one nearest below n_rows / 12 (2048 at cfg1's 25 600 rows; MHR_ROWS_BUCKET overrides)."""
    if bucket is None:
        env = os.environ.get("MHR_ROWS_BUCKET")
        if env:
            bucket = int(env)
        else:
            bucket = max(256, 1 << max(0, int(math.log2(max(1, (n_rows or 12 * 1024) / 12.0)))))
    return max(bucket, -(-int(n_valid) // bucket) * bucket)


def base_config(**kw):
    """The reference's `code/overall/ID.yaml` + `code/IDNet/hstu.yaml` defaults that the hot path reads."""
    cfg = dict(
        seed=2020, MAX_ITEM_LIST_LENGTH=10, loss='nce', neg_sample_mode=None, pad_random_sample=True,
        train_batch_size=64, optim_args={'learning_rate': 1e-3, 'weight_decay': 0.0}, scheduler_args={'type': 'cosine', 'warmup': 0.1},
        category_by='item', eval_batch_size=1024, topk=[5, 10, 50, 200], metrics=['Recall', 'NDCG'], shared_metrics=['Entropy'],
        valid_metric='NDCG@10', metric_decimal_place=7, metrics_pred_len_list=[1], model='HSTU', hidden_act='silu',
        enable_relative_attention_bias=True, total_iters=30000, eval_interval=3000, eval_pred_len=1, pred_len=1,
        split_mode='combine', medusa_lambda=0.99, medusa_num_layers=0, num_segment_head=1, num_prior_head=1,
        head_interaction='multiplicative', eval_num_cats=1, neg_sample_by_cat=True, neg_sample_mix_ratio=0, pos_sample_mix_ratio=0,
        prior_switch=None, weighted_prior_loss=True, outlier_user_metrics=None, n_layers=2, n_heads=1, item_embedding_size=64,
        hstu_embedding_size=64, hidden_dropout_prob=0.1, attn_dropout_prob=0.1, num_negatives=8192, dataset='synthetic',
        accumulate_grad=1, stopping_step=10, suppress_history=True,
    )
    cfg.update(kw)
    return cfg


CONFIGS = {
    # BASELINE.json configs[0]: the reference's CPU-runnable case
    "cfg0": dict(item_num=20001, cfg=base_config(MAX_ITEM_LIST_LENGTH=50, pred_len=1, eval_pred_len=1, n_layers=2, n_heads=1,
                                                  item_embedding_size=64, hstu_embedding_size=64, train_batch_size=64,
                                                  eval_batch_size=256, num_negatives=8192, loss='nce', medusa_num_layers=0)),
    # BASELINE.json configs[1]: HSTU Pixel8M seqlen 200, dim 256, 4 prior heads (the metric's configuration)
    "cfg1": dict(item_num=453938, cfg=base_config(MAX_ITEM_LIST_LENGTH=200, pred_len=8, eval_pred_len=8, n_layers=8, n_heads=8,
                                                   item_embedding_size=256, hstu_embedding_size=256, train_batch_size=128,
                                                   eval_batch_size=256, num_negatives=8192, loss='prior', medusa_num_layers=1,
                                                   num_prior_head=4, num_segment_head=1, head_interaction='multiplicative',
                                                   eval_num_cats=4, neg_sample_by_cat=True,
                                                   optim_args={'learning_rate': 1e-4, 'weight_decay': 0.0})),
    # BASELINE.json configs[2]: HSTU MerRec-shaped, seqlen 512, size-4 (D = 1024, 16 layers x 16 heads), 8 event-type prior
    # heads on one shared negative pool, prior given at test (the catalog size 2^20 is an assumption)
    "cfg2": dict(item_num=1 << 20, cfg=base_config(MAX_ITEM_LIST_LENGTH=512, pred_len=1, eval_pred_len=1, n_layers=16, n_heads=16,
                                                    item_embedding_size=1024, hstu_embedding_size=1024, train_batch_size=64,
                                                    eval_batch_size=64, num_negatives=4096, loss='prior', medusa_num_layers=1,
                                                    num_prior_head=8, num_segment_head=1, head_interaction='multiplicative',
                                                    eval_num_cats=8, neg_sample_by_cat=False, category_by='event',
                                                    prior_given_at_test=True, dataset='merrec-synthetic',
                                                    optim_args={'learning_rate': 1e-4, 'weight_decay': 0.0})),
}
