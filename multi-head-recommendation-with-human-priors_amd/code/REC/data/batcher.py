"""Device-side train / eval batchers with the reference's tensor contract.

The reference builds every sample in Python inside DataLoader workers - `SEQTrainDataset.__getitem__`
(`REC/data/dataset/trainset.py:155-177`: one `process_item` per position, `set` / `np.isin` per negative draw, 70-97),
`SeqEvalDataset.__getitem__` (`evalset.py:80-150`) and `seq_eval_collate` (`collate_fn.py:59-90`) - which becomes the
bottleneck once the GPU step takes 11 ms.  Here a whole batch is produced by a handful of tensor ops on the device that
holds the interaction data (the same code runs on CPU tensors: no kernels of ours are involved), from a CSR copy of the
user sequences and the item -> category multi-hot:

  train batch = (items [B,L+P] i64, neg_items [B,C+1 | 1,n_neg] i64, mask [B,L+P] i64 {0,1}, tags [B,L+P,C] i64 | [B,0])
  eval batch  = (user_ids, item_seq [B,L] front zero-padded, item_target [B,E], (history_u, history_i), positive_u,
                 time_seq, target_tags [B,E,C], outlier_users)

Same distributions as the reference:
  * windows: `valid_sample_locations` as built by `dataload.py:163-195` (one window per short user, non-overlapping
    windows for long users), context front-padded, prediction part end-padded; with `pad_random_sample` the pad slots
    hold uniformly drawn real ids that are not in the user's window (mask 0), else 0;
  * negatives: per category pool (and the global pool last), uniform WITHOUT replacement among the pool's items that
    are not in the padded window (`_neg_sample`; the reference's rare unchecked top-up draw is not reproduced:
    here every negative respects the blacklist); `neg_sample_mix_ratio` flips one coin per (sample, category);
  * sharding: train - a shared permutation per epoch, rank r takes positions r, r+W, ... (DistributedSampler);
    eval - users r, r+W, ... in order (`NonConsecutiveSequentialDistributedSampler`, data/utils.py:95-121).
  * categories (`category_by`): 'item' - the item's multi-hot tags per position (`process_item`); 'user' - the user's
    cluster one-hot on EVERY position of the window, pads included (trainset.py:168-169, evalset.py:119-121); 'event' - the
    one-hot of the interaction's event type on the real positions, zeros on the pads (`process_event`, 143-153; eval:
    the target interactions' event types, evalset.py:144-149, outlier users by unseen target events 126-141).  Per-category
    negative pools exist for 'item' only (trainset.py:33);
  * weighted negatives (`neg_sample_mode`, trainset.py:99-107): draws WITH replacement from the pool with the weight list
    the reference hands to `random.choices(..., weights=...)`, items of the window rejected, repeats allowed.
Out of scope here (data layer): parquet loading, id remapping, timestamps (`time_seq` is returned empty), building the
weight lists from interaction counts (dataload.py:263-327) and the user clustering.
"""
import math

import torch


class SeqStore:
    """User sequences as CSR + item tags, on one device.  `user_seq[uid]` are item ids (uid 0 = padding user, unused),
    `train_seq_len[uid]` the length of the training prefix (dataload.py: the rest is held out), `item_tags` [N, C] bool
    (row 0 = pad item, all False)."""

    def __init__(self, user_seq, train_seq_len, item_tags, device="cpu", user_cluster=None, event_seq=None, item_weights=None,
                 item_weights_by_cat=None, item_raw_tag=None):
        """user_cluster [U] ints (category_by = 'user': `dataload.user_cluster_list`); event_seq: per user, the event type of
        every interaction (category_by = 'event': `dataload.event_seq`, parallel to user_seq); item_weights [N] /
        item_weights_by_cat (list of per-category weight vectors parallel to the category's item list): the lists the
        reference passes as `weights=` when `neg_sample_mode` is set (`dataload.item_interact_weights`, `item_weights_by_cat`)."""
        lens = torch.tensor([len(s) for s in user_seq], dtype=torch.int64)
        self.ptr = torch.zeros(len(user_seq) + 1, dtype=torch.int64)
        self.ptr[1:] = torch.cumsum(lens, 0)
        self.items = torch.tensor([i for s in user_seq for i in s], dtype=torch.int64).to(device)
        self.ptr = self.ptr.to(device)
        self.lens = lens.to(device)
        self.train_len = torch.as_tensor(train_seq_len, dtype=torch.int64).to(device)
        self.item_tags = torch.as_tensor(item_tags).bool().to(device)
        self.item_num = self.item_tags.shape[0]
        self.user_num = len(user_seq)
        self.device = torch.device(device)
        self.user_cluster = None if user_cluster is None else torch.as_tensor(user_cluster, dtype=torch.int64).to(device)
        self.events = None
        if event_seq is not None:
            if [len(e) for e in event_seq] != [len(s_) for s_ in user_seq]:
                raise ValueError("event_seq must be parallel to user_seq")
            self.events = torch.tensor([e for es in event_seq for e in es], dtype=torch.int64).to(device)
        # item_raw_tag [N] ints, -1 = the item has no tag (`dataload.item_to_info[i]['tag']`; outlier_user_metrics = 'tag')
        self.item_raw_tag = None if item_raw_tag is None else torch.as_tensor(item_raw_tag, dtype=torch.int64).to(device)
        self.item_weights = None if item_weights is None else torch.as_tensor(item_weights, dtype=torch.float64).to(device)
        self.item_weights_by_cat = None if item_weights_by_cat is None else [torch.as_tensor(w, dtype=torch.float64).to(device)
                                                                             for w in item_weights_by_cat]

    def sample_locations(self, max_len, pred_len, sample_last_only=False):
        """[n, 2] (uid, context_end) exactly as dataload.py:163-195."""
        out = []
        tl = self.train_len.tolist()
        for uid in range(1, self.user_num):
            n = tl[uid]
            if n <= 1:
                continue
            if sample_last_only:
                out.append((uid, n - 1 if n < pred_len + 3 else n - pred_len))
            elif n <= max_len:
                out.append((uid, n - 1))
            else:
                out += [(uid, e) for e in range((n - 1) % max_len, n, max_len)]
        return torch.tensor(out, dtype=torch.int64).reshape(-1, 2).to(self.device)


def _draw_without_replacement(pool, n_draw, black, gen, slack=16):
    """[B, n_draw] items of `pool` (1-D, unique ids), uniform without replacement per row among the pool items not in
    `black` [B, W] - sequential rejection sampling done in bulk: draw with replacement, drop blacklisted ids and repeats
    (first occurrence wins), keep the first n_draw; rows that come up short are topped up by another round."""
    B, dev = black.shape[0], black.device
    if pool.numel() < n_draw:
        raise ValueError(f"a pool of {pool.numel()} items cannot yield {n_draw} distinct negatives")
    out = torch.zeros(B, n_draw, dtype=torch.int64, device=dev)
    have = torch.zeros(B, dtype=torch.int64, device=dev)
    taken = black
    for _ in range(256):
        need = n_draw - have
        todo = torch.nonzero(need > 0).flatten()
        if todo.numel() == 0:
            return out
        m = int(need.max()) + slack + max(8, int(0.02 * n_draw))
        cand = pool[torch.randint(0, pool.numel(), (todo.numel(), m), device=dev, generator=gen)]
        bl = taken[todo]
        bad = (cand[:, :, None] == bl[:, None, :]).any(-1)
        # repeats inside the draw: keep the first occurrence (stable sort by value, compare neighbours, map back)
        order = torch.argsort(cand, dim=1, stable=True)
        srt = torch.gather(cand, 1, order)
        dup_s = torch.zeros_like(srt, dtype=torch.bool)
        dup_s[:, 1:] = srt[:, 1:] == srt[:, :-1]
        dup = torch.zeros_like(dup_s).scatter_(1, order, dup_s)
        ok = ~(bad | dup)
        rank = torch.cumsum(ok, 1) - 1                                  # position among the accepted ones
        slot = have[todo, None] + rank
        put = ok & (slot < n_draw)
        r_idx = todo[:, None].expand_as(cand)[put]
        out[r_idx, slot[put]] = cand[put]
        got = put.sum(1)
        have[todo] += got
        # accepted ids join the blacklist of their row for the next round (fixed-width: pad with the row's first black id)
        add = torch.where(put, cand, bl[:, :1].expand_as(cand))
        new_taken = taken[:, :1].expand(B, taken.shape[1] + m).clone()
        new_taken[:, :taken.shape[1]] = taken
        new_taken[todo, taken.shape[1]:] = add
        taken = new_taken
    raise ValueError(f"could not draw {n_draw} distinct items outside the blacklist from a pool of {pool.numel()} "
                     "(pool minus window too small)")


def _draw_weighted(pool, weights, n_draw, black, gen):
    """[B, n_draw] items of `pool` drawn WITH replacement with probability proportional to `weights` (parallel to pool), items
    in `black` [B, W] rejected, repeats allowed - the reference's weighted path (trainset.py:99-107: `random.choices` in a
    loop until k accepted)."""
    B, dev = black.shape[0], black.device
    out = torch.zeros(B, n_draw, dtype=torch.int64, device=dev)
    have = torch.zeros(B, dtype=torch.int64, device=dev)
    p = (weights / weights.sum()).float()
    for _ in range(256):
        need = n_draw - have
        todo = torch.nonzero(need > 0).flatten()
        if todo.numel() == 0:
            return out
        m = int(need.max()) + 16
        pick = torch.multinomial(p, todo.numel() * m, replacement=True, generator=gen).view(todo.numel(), m)
        cand = pool[pick]
        ok = ~(cand[:, :, None] == black[todo][:, None, :]).any(-1)
        slot = have[todo, None] + torch.cumsum(ok, 1) - 1
        put = ok & (slot < n_draw)
        out[todo[:, None].expand_as(cand)[put], slot[put]] = cand[put]
        have[todo] += put.sum(1)
    raise ValueError("weighted negative sampling did not find enough items outside the window")


class SeqTrainBatcher:
    """Iterable over train batches for one rank (reference SEQTrainDataset + DistributedSampler + default collate)."""

    def __init__(self, config, store, batch_size=None, rank=0, world=1, seed=2020):
        self.store, self.rank, self.world = store, rank, world
        self.L, self.P = config['MAX_ITEM_LIST_LENGTH'], config['pred_len']
        self.B = batch_size or config['train_batch_size']
        self.return_tags = config['loss'] == 'prior'
        self.category_by = config['category_by'] or 'item'
        if self.category_by not in ('item', 'user', 'event'):
            raise ValueError(f"category_by = {self.category_by} is not defined.")
        self.by_cat = self.return_tags and bool(config['neg_sample_by_cat']) and self.category_by == 'item'
        self.weighted = config['neg_sample_mode'] is not None
        if self.weighted and store.item_weights is None:
            raise ValueError("neg_sample_mode needs SeqStore(item_weights=...) (dataload.item_interact_weights)")
        if self.return_tags and self.category_by == 'user' and store.user_cluster is None:
            raise ValueError("category_by = 'user' needs SeqStore(user_cluster=...)")
        if self.return_tags and self.category_by == 'event' and store.events is None:
            raise ValueError("category_by = 'event' needs SeqStore(event_seq=...)")
        self.n_cats = config['eval_num_cats'] or store.item_tags.shape[1]
        self.mix = config['neg_sample_mix_ratio'] or 0.0
        self.random_pad = bool(config['pad_random_sample'])
        nn_ = config['num_negatives']
        self.n_neg = math.ceil(nn_ / world / self.B) if nn_ else self.L            # trainset.py:60-65
        self.loc = store.sample_locations(self.L, self.P, bool(config['sample_last_only']))
        dev = store.device
        self.all_items = torch.arange(1, store.item_num, device=dev)
        self.C = store.item_tags.shape[1]
        self.cat_items = [torch.nonzero(store.item_tags[:, c]).flatten() for c in range(self.C)] if self.by_cat else []
        self.gen = torch.Generator(device=dev)
        self.gen.manual_seed(seed + 1000003 * (rank + 1))
        self.perm_gen = torch.Generator()                              # shared by all ranks: same permutation per epoch
        self.seed, self.epoch = seed, 0

    def __len__(self):
        return math.ceil(math.ceil(self.loc.shape[0] / self.world) / self.B)

    def batch(self, idx):
        """idx [B] rows of the sample-location table -> the 4-tuple."""
        st, L, P, dev = self.store, self.L, self.P, self.store.device
        W = L + P
        uid, end = self.loc[idx, 0], self.loc[idx, 1]
        start = (end - L).clamp_min(0)
        ctx_pad = L - (end - start)
        n_pred = torch.minimum(st.train_len[uid] - end, torch.full_like(end, P))
        j = torch.arange(W, device=dev)[None, :]
        real = (j >= ctx_pad[:, None]) & (j < (L + n_pred)[:, None])                          # mask 1 positions
        src = st.ptr[uid][:, None] + start[:, None] + (j - ctx_pad[:, None])
        items = torch.where(real, st.items[src.clamp(0, st.items.numel() - 1)], torch.zeros_like(src))
        if self.random_pad:                                     # pad slots: random real ids outside the user's window
            pads = _draw_without_replacement(self.all_items, W, items, self.gen)
            items = torch.where(real, items, pads)
        def draw(pool, weights):
            if self.weighted:
                return _draw_weighted(pool, weights, self.n_neg, items, self.gen)
            return _draw_without_replacement(pool, self.n_neg, items, self.gen)

        w_all = st.item_weights[1:] if self.weighted else None
        pools = []
        if self.by_cat:
            for c in range(self.C):
                neg = draw(self.cat_items[c], st.item_weights_by_cat[c] if (self.weighted and st.item_weights_by_cat) else
                           (st.item_weights[self.cat_items[c]] if self.weighted else None))
                if self.mix > 0.0:                              # one coin per (sample, category): global pool instead
                    glob = draw(self.all_items, w_all)
                    coin = torch.rand(items.shape[0], device=dev, generator=self.gen) <= self.mix
                    neg = torch.where(coin[:, None], glob, neg)
                pools.append(neg)
        pools.append(draw(self.all_items, w_all))
        neg_items = torch.stack(pools, dim=1)
        if not self.return_tags:
            tags = torch.zeros(items.shape[0], 0, dtype=torch.int64, device=dev)
        elif self.category_by == 'item':
            tags = st.item_tags[items].long()
        elif self.category_by == 'user':                        # the user's cluster on every position, pads included
            tags = torch.nn.functional.one_hot(st.user_cluster[uid], self.n_cats)[:, None, :].expand(-1, W, -1).contiguous()
        else:                                                   # the interaction's event type; zeros on the pads
            ev = st.events[src.clamp(0, st.events.numel() - 1)]
            tags = torch.nn.functional.one_hot(ev.clamp(0, self.n_cats - 1), self.n_cats) * real[..., None].long()
        return items, neg_items, real.long(), tags

    def __iter__(self):
        n = self.loc.shape[0]
        self.perm_gen.manual_seed(self.seed + self.epoch)
        self.epoch += 1
        perm = torch.randperm(n, generator=self.perm_gen)
        total = math.ceil(n / self.world) * self.world
        perm = torch.cat([perm, perm[:total - n]])[self.rank::self.world].to(self.store.device)   # DistributedSampler padding
        for i0 in range(0, perm.numel(), self.B):
            yield self.batch(perm[i0:i0 + self.B])


class SeqEvalBatcher:
    """Iterable over eval batches for one rank (reference SeqEvalDataset + seq_eval_collate + rank-strided sampler)."""

    def __init__(self, config, store, phase='valid', batch_size=None, rank=0, world=1):
        self.store, self.phase = store, phase
        self.L = config['MAX_ITEM_LIST_LENGTH_TEST'] or config['MAX_ITEM_LIST_LENGTH']
        self.E = config['eval_pred_len']
        self.B = batch_size or config['eval_batch_size']
        self.return_tags = (config['eval_num_cats'] or 1) > 1
        self.outlier = config['outlier_user_metrics']
        self.category_by = config['category_by'] or 'item'
        self.n_cats = config['eval_num_cats'] or store.item_tags.shape[1]
        if self.return_tags and self.category_by == 'user' and store.user_cluster is None:
            raise ValueError("category_by = 'user' needs SeqStore(user_cluster=...)")
        if self.category_by == 'event' and (self.return_tags or self.outlier == 'event') and store.events is None:
            raise ValueError("category_by = 'event' needs SeqStore(event_seq=...)")
        need = (store.train_len if phase == 'valid' else store.lens - self.E) + self.E
        if bool((need[1:] > store.lens[1:]).any()) or bool(((store.lens - self.E)[1:] < 0).any()):
            raise ValueError("every user needs eval_pred_len held-out items after the history (evalset.py:80-90)")
        self.users = torch.arange(1 + rank, store.user_num, world, device=store.device)
        self.num_total_examples = store.user_num - 1
        self.item_tags = store.item_tags

    def __len__(self):
        return math.ceil(self.users.numel() / self.B)

    def batch(self, uid):
        st, L, E, dev = self.store, self.L, self.E, self.store.device
        B = uid.numel()
        n_hist = st.train_len[uid] if self.phase == 'valid' else st.lens[uid] - E
        base = st.ptr[uid]
        j = torch.arange(L, device=dev)[None, :]
        pos = n_hist[:, None] - L + j                                                          # front zero padding
        item_seq = torch.where(pos >= 0, st.items[(base[:, None] + pos).clamp(0, st.items.numel() - 1)], torch.zeros_like(pos))
        e = torch.arange(E, device=dev)[None, :]
        item_target = st.items[(base[:, None] + n_hist[:, None] + e).clamp(0, st.items.numel() - 1)]
        # full history (not just the last L), as (row, item) pairs
        hu = torch.repeat_interleave(torch.arange(B, device=dev), n_hist)
        off = torch.arange(int(n_hist.sum()), device=dev) - torch.repeat_interleave(torch.cumsum(n_hist, 0) - n_hist, n_hist)
        hi = st.items[torch.repeat_interleave(base, n_hist) + off]
        positive_u = torch.arange(B, device=dev)[:, None].repeat(1, E)
        tgt_pos = (base[:, None] + n_hist[:, None] + e).clamp(0, st.items.numel() - 1)
        if not self.return_tags:
            tt = torch.zeros(B, 0, dtype=torch.int64, device=dev)
        elif self.category_by == 'item':
            tt = st.item_tags[item_target].long()
        elif self.category_by == 'user':
            tt = torch.nn.functional.one_hot(st.user_cluster[uid], self.n_cats)[:, None, :].expand(-1, E, -1).contiguous()
        else:
            tt = torch.nn.functional.one_hot(st.events[tgt_pos].clamp(0, self.n_cats - 1), self.n_cats)
        outlier = torch.zeros(B, dtype=torch.bool, device=dev)
        if self.outlier == 'event' and self.category_by == 'event':     # a target event type the last L history events never showed
            h_ev = torch.where(pos >= 0, st.events[(base[:, None] + pos).clamp(0, st.events.numel() - 1)], torch.full_like(pos, -1))
            outlier = ~(st.events[tgt_pos][:, :, None] == h_ev[:, None, :]).any(-1).all(-1)
        if self.outlier == 'tag' and self.category_by == 'item':        # a target tag the whole history never showed (evalset.py:67-79)
            if st.item_raw_tag is None:
                raise ValueError("outlier_user_metrics = 'tag' needs SeqStore(item_raw_tag=...)")
            n_tag = int(st.item_raw_tag.max()) + 2                       # slot 0: "no tag"
            seen = torch.zeros(B, n_tag, dtype=torch.int32, device=dev)
            seen.index_put_((hu, st.item_raw_tag[hi] + 1), torch.ones_like(hu, dtype=torch.int32), accumulate=True)
            t_tag = st.item_raw_tag[item_target] + 1                     # [B, E]
            outlier = ((t_tag > 0) & (torch.gather(seen, 1, t_tag) == 0)).any(-1)
        if self.outlier == 'category':                        # a target category the whole history never showed (evalset.py:101-109)
            def fixed(t):                                     # items tagged with EVERY category count as untagged
                return t & ~t.all(-1, keepdim=True)
            cnt = torch.zeros(B, st.item_tags.shape[1], dtype=torch.int32, device=dev)
            cnt.index_add_(0, hu, fixed(st.item_tags[hi]).int())
            outlier = (fixed(st.item_tags[item_target]) & ~(cnt > 0)[:, None, :]).any(-1).any(-1)
        return uid, item_seq, item_target, (hu, hi), positive_u, torch.zeros(B, 0, device=dev), tt, outlier

    def __iter__(self):
        for i0 in range(0, self.users.numel(), self.B):
            yield self.batch(self.users[i0:i0 + self.B])
