"""ComiRec baseline on the HSTU encoder, MI355X-native.

Same registry surface, constructor keys and state_dict names as the reference `code/REC/model/IDNet/comirec.py`
(class ComiRec, lines 20-410): the HSTU blocks, item table, `attention_net` (the self-attentive interest extractor) and
`logit_scale`.  Encoder, embedding gather / sparse gradient, sampled softmax and the fused catalog decode are the kernels
of the HSTU path; what ComiRec adds is the multi-interest read-out:

  * training (comirec.py:228-288): K causal interests per position, interest[b,t,k] = sum_{m<=t} softmax_m(a[b,m,k]) out[b,m]
    with a = attention_net(out).  The reference materialises the [B*L, L, D] context windows (13 GB at cfg1); the scores
    depend on m only, so the same numbers come from two prefix sums (exp-weighted numerator and denominator) - O(L).
    The "hard read-out" picks, per (position, offset), the interest with the largest dot product with the target; that
    choice becomes the token's query row of the grouped sampled-softmax launch (a per-step index table instead of the
    static one of the decoding heads).
  * evaluation (comirec.py:346-395): one interest per "head" over the whole sequence -> the fused multi-head decode.
"""
import torch
import torch.nn as nn

from REC.model.IDNet.hstu import HSTU, truncated_normal


class _Overlay:
    """A config view with a few keys pinned (ComiRec has no decoding heads / prior loss)."""

    def __init__(self, base, **fixed):
        self._base, self._fixed = base, fixed

    def __getitem__(self, k):
        return self._fixed[k] if k in self._fixed else self._base[k]

    def get(self, k, default=None):
        v = self[k]
        return default if v is None else v


class ComiRec(HSTU):
    def __init__(self, config, dataload):
        if config['loss'] != 'nce':
            raise NotImplementedError(f"loss={config['loss']} is not supported")
        super().__init__(_Overlay(config, loss='nce', medusa_num_layers=0, num_segment_head=1, num_prior_head=1,
                                  head_interaction='multiplicative', neg_sample_by_cat=False, prior_switch=None,
                                  weighted_prior_loss=False), dataload)
        if config.get('skip_hstu', False):
            raise NotImplementedError("skip_hstu is a debugging switch of the reference and is not built")
        D = self._hstu_embedding_dim
        self.interest_hidden = config.get('interest_hidden', D // 2)
        self.num_interest = K = config.get('interest_num', 4)
        self.attention_net = nn.Sequential(nn.Linear(D, self.interest_hidden, bias=True), nn.Tanh(),
                                           nn.Dropout(self._linear_dropout_rate), nn.Linear(self.interest_hidden, K, bias=False))
        for p in self.attention_net.parameters():
            truncated_normal(p.data, mean=0.0, std=0.02)
        self.medusa_num_heads = K                     # the decode's "heads" are the interests

    def _interest_scores(self, out):
        with torch.autocast(device_type=out.device.type, dtype=torch.bfloat16, enabled=out.is_cuda):
            return self.attention_net(out).float()                                            # [B,L,K]

    _ihn_beta = 0.0                                   # REMI: interest-aware hard negatives (plain sampled softmax here)

    def _causal_interests(self, out, valid, want_stats=False):
        """[B,L,D] fp32, valid [B,L] bool -> [B,L,K,D] fp32 by prefix sums (see the module docstring); want_stats: also the
        routing scores a [B,L,K] (-inf at padded positions; the routing weight of m in window t is softmax_{m <= t}(a)) and the
        running sums den [B,L,K]."""
        a = self._interest_scores(out).masked_fill(~valid[..., None], float("-inf"))
        m = a.max(dim=1, keepdim=True).values
        e = torch.exp(a - torch.where(torch.isfinite(m), m, torch.zeros_like(m)))             # 0 at padded positions
        den = torch.cumsum(e, dim=1)
        num = torch.cumsum(e[..., None] * out[:, :, None, :], dim=1)
        interests = num / den.clamp_min(1e-30)[..., None]
        return (interests, a, den) if want_stats else interests

    def _routing_loss(self, a, den, valid, model_out):
        """Extra loss on the routing weights (REMI's regulariser); ComiRec has none."""
        return 0.0

    def forward(self, interaction):
        from REC.model.basemodel import all_gather_ids
        from REC.model.hstu_functional import EmbeddingGatherFn, L2NormFn
        items, neg_items, user_mask, _ = interaction
        if not items.is_cuda:
            raise RuntimeError("ComiRec.forward runs on the MI355X only (no CPU path); move the batch to the GPU")
        dev = items.device
        B = items.shape[0]
        L, P, D, K = self.max_seq_length, self.pred_len, self._hstu_embedding_dim, self.num_interest
        mask = user_mask.bool()
        self._step_seed += 1
        self._clamp_logit_scale()
        pool_ids = all_gather_ids(neg_items[:, -1].contiguous()).reshape(-1)                   # comirec.py:222-224
        n_item_ids = B * (L + P)
        ids_all = torch.cat([items.reshape(-1), pool_ids]).contiguous()
        self._table_catch_up(ids_all)
        fused_pos = isinstance(self.item_id_proj_tower, nn.Identity)
        rows_items, rows_negs, x = EmbeddingGatherFn.apply(self.item_embedding.weight, self.position_embedding.weight if fused_pos else None,
                                              ids_all, n_item_ids, L, L + P, self)
        if not fused_pos:
            rows_all = self.item_id_proj_tower(torch.cat([rows_items, rows_negs]))
            rows_items, rows_negs = rows_all[:n_item_ids], rows_all[n_item_ids:]
            x = rows_items.view(B, L + P, D)[:, :L] + self.position_embedding.weight[:L][None]
        e_rows = rows_items
        negs = L2NormFn.apply(rows_negs.contiguous()).view(1, -1, D)
        out = self._encode(x, mask[:, :L].to(torch.uint8).contiguous())                        # [B,L,D] fp32
        interests, e_w, den_w = self._causal_interests(out, mask[:, :L], want_stats=True)      # [B,L,K,D]
        # hard read-out (comirec.py:265-288): the interest with the largest dot product with the target of (l, p)
        idx = torch.arange(L, device=dev)[:, None] + 1 + torch.arange(P, device=dev)[None, :]  # [L,P]
        with torch.no_grad():
            g = torch.bmm(interests.reshape(B, L * K, D), e_rows.view(B, L + P, D).transpose(1, 2)).view(B, L, K, L + P)
            sim = torch.gather(g, 3, idx[None, :, None, :].expand(B, L, K, P))                 # [B,L,K,P]
            best = sim.argmax(dim=2)                                                           # [B,L,P]
            b_i = torch.arange(B, device=dev)[:, None, None]
            l_i = torch.arange(L, device=dev)[None, :, None]
            q_all = (((b_i * L + l_i) * K) + best).reshape(1, -1).int().contiguous()
        base_valid = (mask[:, :L, None] & mask[:, idx])[None]                                   # [1,B,L,P]
        mean_gp, logs = self._grouped_loss(interests.reshape(-1, D), e_rows, negs.contiguous(), base_valid,
                                           torch.zeros(1, P, dtype=torch.long), 0, mask.reshape(1, -1), q_all=q_all,
                                           ihn_beta=self._ihn_beta)
        model_out = dict(logs)
        extra = self._routing_loss(e_w, den_w, mask[:, :L], model_out)
        model_out["loss"] = extra + (mean_gp[0] * self.horizon_discount.float()).sum()
        return model_out

    @torch.no_grad()
    def _interest_heads(self, item_seq):
        """[B,L] ids (front zero padded) -> L2-normalised interests [B,K,D] fp32 (comirec.py:330-384)."""
        from mhr_amd import ops
        self.sync_table()                                   # lazy table optimizer: a read outside the training forward
        B, L = item_seq.shape
        if isinstance(self.item_id_proj_tower, nn.Identity):
            _, x = ops.embedding_gather(self.item_embedding.weight, item_seq.contiguous(), torch.float32,
                                        self.position_embedding.weight, L, torch.float32, want_rows=False)
        else:
            rows, _ = ops.embedding_gather(self.item_embedding.weight, item_seq.contiguous(), torch.float32)
            x = self.item_id_proj_tower(rows) + self.position_embedding.weight[:L][None]
        valid = item_seq != 0
        out = self._encode(x, valid.to(torch.uint8).contiguous(), training=False)
        a = self._interest_scores(out).permute(0, 2, 1).masked_fill(~valid[:, None, :], float("-inf"))
        heads = torch.softmax(a, dim=-1).nan_to_num(0.0) @ out                                  # [B,K,D]
        return ops.l2norm_rows(heads.contiguous(), torch.float32)

    @torch.no_grad()
    def predict_topk(self, item_seq, all_item_feature, all_item_tags, target_tags, history=None, k=200,
                     suppress_history=True, stats=None):
        return self._decode_topk(None, all_item_feature, None, target_tags, history, k, suppress_history, stats, self.item_num,
                                 heads_n=self._interest_heads(item_seq))

    @torch.no_grad()
    def predict(self, item_seq, time_seq, all_item_feature, all_item_tags, target_tags, save_for_eval=False):
        return self._dense_scores(None, all_item_feature, None, target_tags, save_for_eval, heads_n=self._interest_heads(item_seq))
