"""Multi-head prior-guided decoding shared by the sequence encoders: heads, sampled-softmax training loss, prior
switch, fused catalog decode.

The reference carries this code twice, line for line: `model/IDNet/hstu.py:436-572, 648-872, 915-1016` (HSTU) and
`model/HLLM/hllm.py:112-252, 506-763, 785-883` (HLLM).  Here it is one mixin; a model supplies the encoder output
[B,L,D], the target-item rows and the L2-normalised negative pools, and keeps the reference's attribute / state_dict
names (`medusa_head`, `medusa_cat_head`, `medusa_seg_head`, `segment_emb`, `aux_cat_head`, `logit_scale`,
`horizon_discount`).  All arithmetic between the library GEMMs runs in the gfx950 kernels (mhr_amd.ops).
"""
import math
from collections import defaultdict

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

from REC.model.llm_heads import ResBlock


class FusedTopK:
    """Per-head top-k of one eval batch: values [B,H,K] fp32, indices [B,H,K] int64 (value desc, index asc)."""
    __slots__ = ("values", "indices", "logs")

    def __init__(self, values, indices, logs=None):
        self.values, self.indices, self.logs = values, indices, logs


class MultiHeadDecoding:
    """Mixin for an nn.Module.  The host class sets `max_seq_length` and calls `_init_multihead(config, dataload, D,
    temp_init)` once, before its own parameter initialisation."""

    # ------------------------------------------------------------------------------------------
    # construction (reference hstu.py:353-367, 416-572 / hllm.py:74-252)
    # ------------------------------------------------------------------------------------------
    def _init_multihead(self, config, dataload, D, temp_init):
        self._head_dim = D
        self.pred_len = config['pred_len']
        self.medusa_lambda = config['medusa_lambda']
        self.num_segment_head = config['num_segment_head']
        self.num_prior_head = config['num_prior_head']
        self.head_interaction = config['head_interaction']
        if self.head_interaction in ('multiplicative', 'hierarchical'):
            self.medusa_num_heads = self.num_segment_head * self.num_prior_head
        elif self.head_interaction == 'additive':
            self.medusa_num_heads = self.num_segment_head + self.num_prior_head
        else:
            raise ValueError(f'Unknown head_interaction: {config["head_interaction"]}')
        self.medusa_num_layers = nl = config['medusa_num_layers']
        self.category_by = config['category_by']

        self.loss = config['loss']
        self.neg_sample_by_cat = bool(config['neg_sample_by_cat']) and self.loss == 'prior'
        self.pos_sample_mix_ratio = config['pos_sample_mix_ratio'] or 0.0
        if self.loss not in ('nce', 'prior'):
            raise NotImplementedError(f"loss={self.loss} is not supported")
        if config['fix_temp']:
            self.register_buffer("logit_scale", torch.tensor(np.log(temp_init), dtype=torch.float32))
        else:
            self.logit_scale = nn.Parameter(torch.ones([]) * np.log(temp_init))
        self.nce_thres = config['nce_thres'] if config['nce_thres'] else 0.99
        # evaluation ranks fp32 scores of fp32 operands like the reference (bit-exact top-k indices wherever the reference's
        # scores are untied); False: rank the bf16 MFMA scores directly (the candidates' order within 2^-8 may differ)
        self.exact_fp32_topk = bool(config.get('exact_fp32_topk', True))
        self.seg_len = self.pred_len
        if nl > 0:
            assert self.pred_len % self.num_segment_head == 0, "pred_len must be divisible by the number of segments"
            self.seg_len = self.pred_len // self.num_segment_head

        lam = torch.tensor([self.medusa_lambda ** p for p in range(self.pred_len)])
        self.register_buffer('horizon_discount', lam / lam.sum())

        S, C = self.num_segment_head, self.num_prior_head
        if nl == 0:
            self.medusa_head = nn.ModuleList([nn.Identity() for _ in range(self.medusa_num_heads)])
        elif self.head_interaction == 'hierarchical':
            self.head_norm = config.get("head_norm", False)
            self.cat_bottleneck = config.get("cat_bottleneck", False)
            self.cat_bottleneck_dim = config.get("cat_bottleneck_dim", D // 2)
            self.share_seg_weights = config.get("share_seg_weights", False)
            self.use_seg_embed = config.get("segment_embed", False)
            if self.use_seg_embed:
                self.segment_emb = nn.Embedding(S, D)

            def cat_block():
                layers = []
                if self.cat_bottleneck:
                    layers += [nn.LayerNorm(D), nn.Linear(D, self.cat_bottleneck_dim), nn.SiLU(),
                               nn.Linear(self.cat_bottleneck_dim, D)]
                layers += [ResBlock(D, use_norm=self.head_norm, zero_init=False) for _ in range(nl)]
                return nn.Sequential(*layers)

            def seg_block():
                return nn.Sequential(*[ResBlock(D, use_norm=self.head_norm, zero_init=False) for _ in range(nl)])

            self.medusa_cat_head = nn.ModuleList([cat_block() for _ in range(C)])
            if self.share_seg_weights:
                shared = seg_block()
                self.medusa_seg_head = nn.ModuleList([nn.ModuleList([shared for _ in range(S)]) for _ in range(C)])
            else:
                self.medusa_seg_head = nn.ModuleList([nn.ModuleList([seg_block() for _ in range(S)]) for _ in range(C)])
        else:
            # one ResBlock instance applied `nl` times per head (state_dict keys .0. ... alias it), as in the reference
            self.medusa_head = nn.ModuleList([nn.Sequential(*([ResBlock(D)] * nl)) for _ in range(self.medusa_num_heads)])
        if nl > 0 and self.loss != 'prior':
            assert C == 1, 'Only prior loss is allowed for num_prior_head > 1'

        self.weighted_prior_loss = config['weighted_prior_loss']
        if self.loss == 'prior' and self.weighted_prior_loss and nl > 0:
            total = sum(dataload.category_counts.values())
            self.prior_loss_weight = [0.0] * C
            for name, cnt in dataload.category_counts.items():
                self.prior_loss_weight[dataload.category_to_int[name]] = cnt / total
        else:
            self.prior_loss_weight = [1.0 / C] * C
        # prior switch (reference hstu.py:512-544 / hllm.py:209-245): one auxiliary "does category c occur in the next
        # pred_len items" logit per prior head, trained with BCE / asymmetric loss, usable to switch heads off at test time
        self.prior_switch = None
        if self.loss == 'prior' and config['prior_switch'] is not None and nl > 0:
            self.use_asym_switch_loss = config.get('asym_switch_loss', False)
            self.switch_last_only = config.get('switch_last_only', False)
            self.asl_gamma_pos, self.asl_gamma_neg = config.get('gamma_pos', 4.0), config.get('gamma_neg', 0.0)
            assert config['split_mode'] == 'combine'
            self.master_switch = config.get('master_switch', False)
            if config['prior_switch'] == 'in':
                self.aux_cat_head = nn.ModuleList([nn.Linear(D, 1) for _ in range(C)])
                self.prior_switch = 'in'
            elif config['prior_switch'] == 'in_out':
                if self.head_interaction == 'multiplicative':
                    assert S == 1, 'multiplicative head interaction is not supported for prior_switch=in_out when num_segment_head > 1'
                self.aux_cat_head = nn.ModuleList([nn.Linear(2 * D, 1) for _ in range(C)])
                self.prior_switch = 'in_out'
            if self.prior_switch is not None:
                if self.master_switch:
                    for i in range(1, C):
                        for p_ in self.aux_cat_head[i].parameters():
                            p_.requires_grad_(False)
                self.prior_switch_loss_weight = config['prior_switch_loss_weight']
        self.use_prior_switch_test = config.get('use_prior_switch_test', False)
        self.detach_aux_in = config.get('detach_aux_in', False)
        self.eval_pred_len = config['eval_pred_len']
        self.prior_given_at_test = config.get('prior_given_at_test', False)
        self.given_prior_len = config.get('given_prior_len', self.eval_pred_len) if self.prior_given_at_test else self.eval_pred_len
        self.int_to_category = config["int_to_category"]
        self.rank = torch.distributed.get_rank() if torch.distributed.is_initialized() else 0

        # runtime state of the fused path
        self._tok_cache = {}
        self._item_cache = None
        self._row_bits_cache = {}

    # ------------------------------------------------------------------------------------------
    # heads and prior switch
    # ------------------------------------------------------------------------------------------
    def adjacent_parameters(self):
        """Parameter lists the fused optimizer should lay out back to back (mhr_amd.optim.FusedAdamW): the one-ResBlock
        decoding heads' weights, and their biases, which `_heads` uses as ONE concatenated GEMM."""
        heads = getattr(self, 'medusa_head', None)
        if (heads is None or self.medusa_num_layers != 1 or self.head_interaction == 'hierarchical'
                or not all(isinstance(h, nn.Sequential) and len(h) == 1 and hasattr(h[0], "linear") for h in heads)):
            return []
        ws, bs = [h[0].linear.weight for h in heads], [h[0].linear.bias for h in heads]
        if len({id(p) for p in ws}) != len(ws):               # heads sharing one module (tied): nothing to lay out
            return []
        return [ws, bs]

    def _heads(self, x, x_bf16=None):
        """x [..., D] fp32 -> [..., H, D] fp32 (reference hstu.py:652-667, 915-931 / hllm.py:510-525, 785-801).  bf16
        GEMMs, fp32 residual, like the reference under bf16-mixed autocast.  x_bf16: the caller's bf16 copy of x, if it has one."""
        S, C = self.num_segment_head, self.num_prior_head
        heads = getattr(self, 'medusa_head', None)
        if (x.is_cuda and heads is not None and self.medusa_num_layers == 1 and self.head_interaction != 'hierarchical'
                and all(isinstance(h, nn.Sequential) and len(h) == 1 and not h[0].use_norm for h in heads)):
            # one-ResBlock heads: all H projections as ONE GEMM on concatenated weights (x is read once, the weight
            # gradient is one split-K GEMM), SiLU + residual in two elementwise kernels, no stack
            from REC.model.hstu_functional import FusedHeadsLinearFn, SplitKLinearFn
            from mhr_amd import optim as _optim
            H, D = len(heads), x.shape[-1]
            ws, bs = [h[0].linear.weight for h in heads], [h[0].linear.bias for h in heads]
            fw = _optim.fused_views(ws) if torch.is_grad_enabled() else None
            fb = _optim.fused_flat(bs) if fw is not None else None
            xb = x.reshape(-1, D).to(torch.bfloat16) if x_bf16 is None else x_bf16.reshape(-1, D)
            if fw is not None and fb is not None:     # the fused optimizer laid the heads out back to back: views, no cat / cast
                z = FusedHeadsLinearFn.apply(xb, fw[0], fb[0], fw[1], fb[1])
            else:
                w = torch.cat(ws, 0)
                b = torch.cat(bs, 0)
                z = SplitKLinearFn.apply(xb, w, b, True, None)
            if x.dim() == 3 and x.dtype == torch.float32 and D % 4 == 0:
                # bias-added GEMM output -> SiLU + residual written straight in the [B, H, L, D] layout of the loss
                # (csrc/heads.hip); returned as the [B, L, H, D] view every caller expects
                from REC.model.hstu_functional import HeadsResidualFn
                Bq, Lq = x.shape[0], x.shape[1]
                return HeadsResidualFn.apply(x.reshape(-1, D), z, Bq, Lq, H).permute(0, 2, 1, 3)
            return torch.add(x.unsqueeze(-2), F.silu(z.view(*x.shape[:-1], H, D)))
        with torch.autocast(device_type=x.device.type, dtype=torch.bfloat16, enabled=x.is_cuda):
            if self.medusa_num_layers > 0 and self.head_interaction == 'hierarchical':
                cat = [self.medusa_cat_head[c](x) for c in range(C)]
                outs = []
                for s in range(S):
                    bias = self.segment_emb.weight[s] if self.use_seg_embed else None
                    for c in range(C):
                        h_in = cat[c] if bias is None else cat[c] + bias
                        outs.append(self.medusa_seg_head[c][s](h_in))
            else:
                outs = [head(x) for head in self.medusa_head]
        return torch.stack([o.float() for o in outs], dim=-2)

    def _switch_logits(self, out, head_out, c):
        """aux_cat_head[c] on the encoder output ('in') or [encoder output | head c output] ('in_out'); out [..., D],
        head_out [..., H, D] (reference hstu.py:770-784, 940-947).  A D -> 1 GEMV: library matmul, bf16 like the
        reference under autocast."""
        if self.prior_switch == 'in':
            aux_in = out
        else:
            h = self.num_segment_head + c if self.head_interaction == 'additive' else c
            aux_in = torch.cat([out, head_out[..., h, :]], dim=-1)
        if self.detach_aux_in:
            aux_in = aux_in.detach()
        with torch.autocast(device_type=out.device.type, dtype=torch.bfloat16, enabled=out.is_cuda):
            return self.aux_cat_head[c](aux_in).squeeze(-1).float()

    def _switch_loss(self, out, head_out, tag_win, c):
        """(loss, accuracy) of the category-c switch (reference hstu.py:757-805).  tag_win [B,L,P,C] future tags."""
        target = tag_win[..., c].any(dim=2).float()                            # [B,L]
        logits = self._switch_logits(out, head_out, c)                         # [B,L]
        if self.switch_last_only:
            target, logits = target[:, -1:], logits[:, -1:]
        if self.use_asym_switch_loss:      # REC/model/layers.py:16-83, sum over the last dim, mean over the rest
            xs_pos = torch.sigmoid(logits)
            xs_neg = (1 - xs_pos + 0.05).clamp(max=1)
            ls = target * torch.log(xs_pos.clamp(min=1e-8)) + (1 - target) * torch.log(xs_neg.clamp(min=1e-8))
            if self.asl_gamma_neg > 0 or self.asl_gamma_pos > 0:
                pt = xs_pos * target + xs_neg * (1 - target)
                ls = ls * torch.pow(1 - pt, self.asl_gamma_pos * target + self.asl_gamma_neg * (1 - target))
            loss = (-ls.sum(dim=-1)).mean()
        else:
            p = max(min(float(self.prior_loss_weight[c]), 1.0 - 1e-6), 1e-6)
            loss = F.binary_cross_entropy_with_logits(logits, target, pos_weight=torch.tensor((1.0 - p) / p, device=logits.device))
        acc = ((logits >= 0).int() == target.int()).float().mean()
        return loss, acc.detach()

    # ------------------------------------------------------------------------------------------
    # training loss
    # ------------------------------------------------------------------------------------------
    def _const(self, values, device, dtype):
        """Small constant vector on the device, uploaded once (a `torch.tensor(list, device=...)` per step is a blocking copy)."""
        key = ("const", tuple(values), str(device), dtype)
        if key not in self._tok_cache:
            self._tok_cache[key] = torch.tensor(values, device=device, dtype=dtype)
        return self._tok_cache[key]

    def _token_tables(self, B, device, head_for_p):
        """Static index tables for G groups: query row in head_embs.view(-1,D) [G, B*P*L], target row in
        e.view(-1,D) [B*P*L], prediction offset [B*P*L].  head_for_p: [G, P] long (head used by group g at offset p)."""
        key = (B, str(device), tuple(head_for_p.reshape(-1).tolist()))
        if key not in self._tok_cache:
            # slot order (b, l, p), offset fastest: the tokens that share a head row (b, head, l) are neighbours in the
            # compacted lists, which lets the row-wise backward kernel combine their gradient rows before its atomics
            L, P, H = self.max_seq_length, self.pred_len, self.medusa_num_heads
            b = torch.arange(B, device=device)[None, :, None, None]
            l = torch.arange(L, device=device)[None, None, :, None]
            p = torch.arange(P, device=device)[None, None, None, :]
            hp = head_for_p.to(device)[:, None, None, :]                                        # [G,1,1,P]
            G = hp.shape[0]
            q_all = ((b * H + hp) * L + l).expand(G, B, L, P).reshape(G, -1).int().contiguous()
            p_all = (b * (L + P) + l + 1 + p).expand(1, B, L, P).reshape(-1).int().contiguous()
            o_all = p.expand(1, B, L, P).reshape(-1).int().contiguous()
            self._tok_cache[key] = (q_all, p_all, o_all)
        return self._tok_cache[key]

    def _groups_read_disjoint_heads(self, head_for_p):
        """head_for_p [G, P] (host): True when no decoding head is read by two groups - then every query row (b, head, l) belongs
        to exactly one (group, row) pair of the row-shared loss and its gradient needs no atomics."""
        key = ("disjoint", tuple(head_for_p.reshape(-1).tolist()), tuple(head_for_p.shape))
        if key not in self._tok_cache:
            sets = [set(row.tolist()) for row in head_for_p]
            self._tok_cache[key] = all(a.isdisjoint(b) for i, a in enumerate(sets) for b in sets[i + 1:])
        return self._tok_cache[key]

    def _token_lists(self, valid_g, head_for_p, q_all=None):
        """Ordered compaction of the live (group, slot) pairs of valid_g [G,B,L,P]: (share, q_idx, p_idx, o_idx, n_tok, window)."""
        from mhr_amd import ops
        G, B, L, P = valid_g.shape
        n_slots = B * P * L
        q_static, p_all, o_all = self._token_tables(B, valid_g.device, head_for_p)
        q_all = q_static if q_all is None else q_all          # (a model may pick the query row per token itself: ComiRec)
        share = P > 1 and q_all is q_static      # the P offsets of a position share their query row (csrc/nce_shared.hip)
        if share:
            q_idx, p_idx, o_idx, n_tok, tos = ops.token_compact(valid_g.reshape(G, n_slots), q_all, p_all, o_all, slot_map=True)
            return share, q_idx, p_idx, o_idx, n_tok, (tos, L, P)
        q_idx, p_idx, o_idx, n_tok = ops.token_compact(valid_g.reshape(G, n_slots), q_all, p_all, o_all)
        return share, q_idx, p_idx, o_idx, n_tok, None

    def _early_loss_stages(self, st, pools, mask, pos_tags, e_rows, negs_pools, n_q_rows=None):
        """The batch-only half of the grouped loss as a list of stages (callables, run in order on ONE stream - HSTU.forward
        runs them on a second stream, one between every two encoder layers): the loss plan, the token lists of its groups and
        the stages of `ops.nce_shared_prepare_stages` (row maps, false-negative bit table, normalised targets, accumulators) -
        some forty small launches and one MFMA kernel that wait for nothing the encoder makes.  Results land in `st`:
        st["plan"], st["early"] = dict(tokens, prep)."""
        from mhr_amd import ops
        stages = []

        def plan():
            st["plan"] = self._loss_plan(None, pools, mask, pos_tags)

        def tokens():
            pl = st["plan"]
            tk = self._token_lists(pl["valid_g"], pl["head_for_p_g"])
            st["early"] = dict(tokens=tk, prep=None)
            if tk[0] and pl["slots"] == list(range(negs_pools.shape[0])):          # negs_g IS negs_pools: same memory on both streams
                prep, more = ops.nce_shared_prepare_stages(tk[1], tk[2], tk[4], e_rows.detach(), negs_pools.detach().contiguous(),
                                                           float(self.nce_thres), pl["p_row_mask"], pl["log_group"] is not None,
                                                           n_q_rows=n_q_rows)
                st["early"]["prep"] = prep
                stages[0:0] = more                # next in line (this stage has been taken off the list)

        stages += [plan, tokens]
        return stages

    def _grouped_loss(self, head_rows, e_rows, negs_g, valid_g, head_for_p, log_group, p_row_mask=None, q_all=None, ihn_beta=0.0,
                      early=None, bucket_weight=None, n_segments=1):
        """Sampled-softmax loss of G (token mask, negative pool) groups in ONE fused launch per kernel.
        valid_g [G,B,L,P] bool (offset fastest), negs_g [G,n_neg,D] bf16, head_for_p [G,P].  Returns (mean loss per (group, offset)
        [G,P] fp32, logs of `log_group` or None).  No host sync: tokens are compacted by scatter at fixed capacity
        and the live counts stay on the device (reference: boolean-mask compaction + `mask.sum() == 0` host branch,
        hstu.py:688-690, 814-829)."""
        from REC.model.hstu_functional import NceLossFn
        G, B, L, P = valid_g.shape
        from mhr_amd import ops
        share, q_idx, p_idx, o_idx, n_tok, window = early["tokens"] if early is not None else self._token_lists(valid_g, head_for_p, q_all)
        cap = q_idx.shape[1]
        want_logs = log_group is not None
        logs = {} if want_logs else None
        mean_p = NceLossFn.apply(head_rows, e_rows, negs_g, self.logit_scale, q_idx, p_idx, n_tok, cap,
                                 float(self.nce_thres), want_logs, logs, o_idx, P,
                                 log_group if want_logs else -1, p_row_mask, share, window, float(ihn_beta),
                                 early["prep"] if early is not None else None, bucket_weight, n_segments,
                                 share and self._groups_read_disjoint_heads(head_for_p))
        # [G, P] mean loss per (group, offset) - or, with bucket_weight [G, P], the pair (sum of weight x mean, flat log vector
        # per_gp | seg_all | g_tot | seg_sum) straight from one launch (ops.loss_reduce)
        out_logs = None
        if want_logs:                                   # all counters of the logged group in ONE launch (mhr_nce_log_counters)
            ks = [k for k in (1, 5, 10, 50, 100) if k <= negs_g.shape[1] + 1]
            vals = ops.nce_log_counters(logs["n_valid"], logs["rank"], o_idx, n_tok, log_group, ks)       # [1 + len(ks)]
            out_logs = {'nce_samples': vals[0]}
            for i, k in enumerate(ks):
                out_logs[f'nce_top{k}_acc'] = vals[1 + i]
        return mean_p, out_logs

    def _clamp_logit_scale(self):
        if isinstance(self.logit_scale, nn.Parameter):
            with torch.no_grad():
                self.logit_scale.clamp_(0, math.log(100))                # reference hstu.py:601-602 (in place)

    def _negative_pools(self, n_pools):
        """Indices of the negative pools the loss reads, in the order `_multihead_loss` expects them (reference
        hstu.py:669-670, 751-752): the shared pool (last one) for nce / additive, then one per prior category."""
        pools = []
        if (not self.neg_sample_by_cat) or (self.loss == 'prior' and self.head_interaction == 'additive'):
            pools.append(n_pools - 1)
        if self.loss == 'prior' and self.neg_sample_by_cat:
            pools += list(range(self.num_prior_head))
        return pools

    def _loss_plan(self, negs_pools, pools, mask, pos_tags):
        """Everything of the loss that depends on the BATCH only, not on the encoder output: the token masks of the groups
        (one group per (token mask, head assignment, negative pool): the nce branch and every prior category), their heads /
        pools / weights, the rows of the target table each group can point at."""
        dev = mask.device
        L, P = self.max_seq_length, self.pred_len
        S, C = self.num_segment_head, self.num_prior_head
        additive = self.head_interaction == 'additive'
        pool_slot = {p: i for i, p in enumerate(pools)}
        # the window of (l, p) is position l + 1 + p: a strided VIEW of the row (no index kernel, no [L, P] index tensor)
        base_valid = mask[:, :L, None] & mask[:, 1:].unfold(1, P, 1)                               # [B,L,P]
        groups = []                       # (head_for_p [P], pool slot, weight, kind, index)
        valid_parts, row_parts = [], []   # token masks [*,B,L,P] and, per group, the target rows (b, j) its tokens can point at
        if self.loss == 'nce' or (self.loss == 'prior' and additive):
            groups.append((torch.arange(P) // self.seg_len, pool_slot[pools[0]], 1.0, 'nce', 0))
            valid_parts.append(base_valid[None])
            row_parts.append(mask[None])
        tag_win = None
        if self.loss == 'prior':
            seg_len = P if additive else self.seg_len
            seg_for_p = torch.arange(P) // seg_len
            tags_b = pos_tags.bool()                                                               # [B,L+P,C]
            tag_win = tags_b[:, 1:].unfold(1, P, 1).permute(0, 1, 3, 2)                            # [B,L,P,C] (a view)
            tag_cblp = tag_win.permute(3, 0, 1, 2)                                                 # [C,B,L,P] (a view)
            if self.pos_sample_mix_ratio > 0.0:
                tag_cblp = tag_cblp | (torch.rand(tag_cblp.shape, device=dev) < self.pos_sample_mix_ratio)
                row_parts.append(mask[None].expand(C, -1, -1))
            else:
                row_parts.append(mask[None] & tags_b.permute(2, 0, 1))
            valid_parts.append(base_valid[None] & tag_cblp)                                        # all C categories in one pass
            for c in range(C):
                head_for_p = torch.full((P,), S + c) if additive else seg_for_p * C + c
                pool = pool_slot[c] if self.neg_sample_by_cat else pool_slot[pools[0]]
                groups.append((head_for_p, pool, float(self.prior_loss_weight[c]), 'prior', c))
        slots = [g[1] for g in groups]
        negs_g = None                     # (negs_pools=None: the caller picks the pools itself, `_multihead_loss` does)
        if negs_pools is not None:
            negs_g = (negs_pools if slots == list(range(len(pools))) else negs_pools[self._const(slots, dev, torch.int64)]).contiguous()
        valid_g = valid_parts[0] if len(valid_parts) == 1 else torch.cat(valid_parts)
        rows_g = row_parts[0] if len(row_parts) == 1 else torch.cat(row_parts)
        return dict(groups=groups, tag_win=tag_win, valid_g=valid_g.contiguous(),
                    head_for_p_g=torch.stack([g[0] for g in groups]), negs_g=negs_g, slots=slots,
                    # reference: top-k logs come from the nce branch, then are overwritten by prior category 0 (hstu.py:723, 863)
                    log_group=max(i for i, g in enumerate(groups) if g[3] == 'nce' or g[4] == 0),
                    p_row_mask=rows_g.reshape(len(groups), -1).contiguous())

    def _multihead_loss(self, out, e_rows, negs_pools, pools, mask, pos_tags, plan=None, early=None, out_bf16=None):
        """Everything of the training forward after the sequence encoder (reference hstu.py:648-872 / hllm.py:506-763).
        out [B,L,D] fp32 encoder output; e_rows [B*(L+P), D] fp32 target-item rows; negs_pools [len(pools), n_pool, D]
        bf16 L2-normalised; mask [B,L+P] bool; pos_tags [B,L+P,C]; plan / early: the results of `_early_loss_stages(...)` when the
        caller built them before the encoder ran."""
        dev = out.device
        B = out.shape[0]
        L, P, D = self.max_seq_length, self.pred_len, self._head_dim
        S, C = self.num_segment_head, self.num_prior_head
        additive = self.head_interaction == 'additive'
        head_embs = self._heads(out, out_bf16).permute(0, 2, 1, 3).contiguous()     # [B,H,L,D]
        head_rows = head_embs.view(-1, D)
        if plan is None:
            plan = self._loss_plan(negs_pools, pools, mask, pos_tags)
        groups, tag_win, valid_g, head_for_p_g = plan["groups"], plan["tag_win"], plan["valid_g"], plan["head_for_p_g"]
        negs_g, log_group, p_row_mask = plan["negs_g"], plan["log_group"], plan["p_row_mask"]
        if negs_g is None:
            slots = plan["slots"]
            negs_g = negs_pools if slots == list(range(len(pools))) else negs_pools[self._const(slots, dev, torch.int64)]
        gw = ("gw", tuple(g[2] for g in groups), str(dev))                 # horizon discount x group weight, [G,P], built once
        if gw not in self._tok_cache:
            self._tok_cache[gw] = (self.horizon_discount.float()[None, :] * self._const([g[2] for g in groups], dev, torch.float32)[:, None]).contiguous()
        G = len(groups)
        model_out = defaultdict(float)
        # the mean per offset, its weighting, the total and the log values (the reference's per-head / per-segment sums,
        # hstu.py:700-721, 836-858) come out of ONE launch behind the loss kernels; the log values are views of its output
        (total, red), logs = self._grouped_loss(head_rows, e_rows, negs_g.contiguous(), valid_g, head_for_p_g, log_group,
                                                p_row_mask, early=early, bucket_weight=self._tok_cache[gw], n_segments=S)
        seg_all = red[G * P:G * P + G * S].view(G, S)
        g_tot = red[G * P + G * S:G * P + G * S + G]
        seg_tot = seg_all[0] if (additive or groups[0][3] == 'nce') else red[G * P + G * S + G:]
        for s_ in range(S):
            model_out[f"seg_{s_}_loss"] = seg_tot[s_]
        for i, g in enumerate(groups):
            if g[3] == 'prior':
                model_out[f'head_nce_{self.int_to_category[g[4]]}_loss'] = g_tot[i]
        if self.loss == 'prior' and additive:
            total = total / 2
        if self.prior_switch is not None:
            head_out = head_embs.permute(0, 2, 1, 3)                               # [B,L,H,D]
            for c in range(1 if self.master_switch else C):
                sw_loss, sw_acc = self._switch_loss(out, head_out, tag_win, c)
                name = self.int_to_category[c]
                model_out[f'head_cat_{name}_acc'] = sw_acc
                model_out[f'head_cat_{name}_loss'] = self.prior_switch_loss_weight * sw_loss.detach()
                # (the reference adds it inside the category loop, i.e. before the additive mode's final halving, hstu.py:870)
                total = total + self.prior_switch_loss_weight * sw_loss * (0.5 if additive else 1.0)
        model_out.update(logs)
        model_out["loss"] = total
        return model_out

    # ------------------------------------------------------------------------------------------
    # evaluation
    # ------------------------------------------------------------------------------------------
    @torch.no_grad()
    def _heads_at_last(self, last, want_switch=False):
        """last [B,D] fp32 (encoder output at the last position) -> L2-normalised head embeddings [B,H,D] fp32 and, with
        want_switch, the prior-switch decisions [B, n] bool, n = 1 under master_switch (reference hstu.py:915-966)."""
        from mhr_amd import ops
        heads = self._heads(last)                                         # [B,H,D]
        if want_switch:
            n = 1 if self.master_switch else self.num_prior_head
            pred = torch.stack([self._switch_logits(last, heads, c) >= 0 for c in range(n)], dim=1)
            return ops.l2norm_rows(heads.contiguous(), torch.float32), pred
        return ops.l2norm_rows(heads.contiguous(), torch.float32)

    def _switched_off(self, pred):
        """[B,C] bool: prior heads the switch turns off (reference hstu.py:1002-1011)."""
        if self.master_switch:
            return torch.cat([~pred[:, :1], pred[:, :1].expand(-1, self.num_prior_head - 1)], dim=1)
        return ~pred

    def _switch_logs(self, pred, target_tags, logs):
        for c in range(pred.shape[1]):
            label = target_tags[:, :, c].sum(dim=-1) > 0
            logs[f'head_cat_{self.int_to_category[c]}_num_correct'] = torch.sum((label == pred[:, c]) * 1.0)

    def _row_constraints(self, B, target_tags, device, switched_off=None):
        """Per (user, head) admissible-category bit (int32 bit pattern): bit c for prior heads, bit 31 for
        unconstrained heads, 0 for heads switched off by `prior_given_at_test` (reference hstu.py:982-999)."""
        S, C, H = self.num_segment_head, self.num_prior_head, self.medusa_num_heads
        static = (self.loss != 'prior' or not self.prior_given_at_test) and switched_off is None
        if static and (B, device) in self._row_bits_cache:
            return self._row_bits_cache[(B, device)]
        if self.loss != 'prior':
            bits = torch.full((B * H,), -(1 << 31), dtype=torch.int32, device=device)
            self._row_bits_cache[(B, device)] = bits
            return bits
        additive = self.head_interaction == 'additive'
        cat_of_head = torch.tensor([(h - S if h >= S else -1) if additive else h % C for h in range(H)], device=device)
        bits = torch.where(cat_of_head >= 0, torch.ones_like(cat_of_head) << cat_of_head.clamp_min(0),
                           torch.full_like(cat_of_head, -(1 << 31)))
        bits = bits[None, :].expand(B, H).clone()
        if self.prior_given_at_test:
            given = target_tags[:, :self.given_prior_len].bool().any(dim=1)                       # [B,C]
            on = torch.where(cat_of_head[None, :] >= 0, given[:, cat_of_head.clamp_min(0)], torch.ones_like(bits, dtype=torch.bool))
            bits = torch.where(on, bits, torch.zeros_like(bits))
        if switched_off is not None:                                                              # hstu.py:1002-1015
            off = torch.where(cat_of_head[None, :] >= 0, switched_off[:, cat_of_head.clamp_min(0)], torch.zeros_like(bits, dtype=torch.bool))
            bits = torch.where(off, torch.zeros_like(bits), bits)
        bits = bits.reshape(-1).to(torch.int32)
        if static:
            self._row_bits_cache[(B, device)] = bits
        return bits

    @staticmethod
    def pack_item_tags(all_item_tags):
        """[C,N] {0,1} -> [N] int32 bit patterns, bit 31 always set (C <= 31)."""
        C, N = all_item_tags.shape
        w = (1 << torch.arange(C, device=all_item_tags.device, dtype=torch.int64))[:, None]
        bits = (all_item_tags.bool().long() * w).sum(0) | (1 << 31)
        return torch.where(bits >= (1 << 31), bits - (1 << 32), bits).to(torch.int32)

    def _normalised_items(self, all_item_feature, all_item_tags):
        """bf16 L2-normalised item table padded to whole 32-row tiles + packed tag words, cached per (table, tags)."""
        from mhr_amd import ops
        key = (all_item_feature.data_ptr(), all_item_feature._version, None if all_item_tags is None else all_item_tags.data_ptr())
        if self._item_cache is None or self._item_cache[0] != key:
            feat = all_item_feature.float().contiguous()
            items_bf = ops.l2norm_rows(feat, torch.bfloat16)                                       # hstu.py:974-975
            if items_bf.shape[0] % 32:      # whole 32-row tiles: the catalog scorer then streams it unclamped
                items_bf = torch.cat([items_bf, items_bf.new_zeros(32 - items_bf.shape[0] % 32, items_bf.shape[1])]).contiguous()
            tag_bits = self.pack_item_tags(all_item_tags) if (all_item_tags is not None and self.loss == 'prior') else None
            # fp32 normalised rows for the exact re-score of the final candidates (the reference ranks fp32 scores)
            items_f32 = ops.l2norm_rows(feat, torch.float32) if self.exact_fp32_topk else None
            self._item_cache = (key, items_bf, tag_bits, items_f32)
        return self._item_cache[1], self._item_cache[2]

    @torch.no_grad()
    def _decode_topk(self, last, all_item_feature, all_item_tags, target_tags, history, k, suppress_history, stats, n_ids,
                     heads_n=None, defer=False):
        """Heads at the last position -> catalog scoring with tag / pad / history / switch masks -> exact per-head top-k.
        Replaces reference hstu.py:915-1015 + trainer.py:724-726 + collector.py:245 without the [B,H,N] tensor."""
        from mhr_amd import ops
        B = (last if heads_n is None else heads_n).shape[0]
        H = self.medusa_num_heads
        dev = (last if heads_n is None else heads_n).device
        switched_off, logs = None, {'num_samples': self.eval_pred_len * B}
        if heads_n is not None:                          # the model brings its own L2-normalised [B,H,D] read-outs
            pass
        elif self.prior_switch is not None:
            heads_n, pred = self._heads_at_last(last, want_switch=True)
            self._switch_logs(pred, target_tags, logs)
            if self.use_prior_switch_test:
                switched_off = self._switched_off(pred)
        else:
            heads_n = self._heads_at_last(last)
        users = heads_n.to(torch.bfloat16).view(B * H, -1).contiguous()
        items_bf, tag_bits = self._normalised_items(all_item_feature, all_item_tags)
        row_bits = self._row_constraints(B, target_tags, dev, switched_off)
        hist_ptr = hist_items = None
        if suppress_history and history is not None and history[0].numel() > 0:
            hu, hi = history[0].to(dev), history[1].to(dev)
            order = torch.argsort(hu * n_ids + hi)
            hist_items = hi[order].contiguous()
            # CSR offsets by binary search on the sorted user column (torch.bincount would sync the host)
            hist_ptr = torch.searchsorted(hu[order].contiguous(), torch.arange(B + 1, device=dev)).int()
        if self.exact_fp32_topk:
            # ranked on fp32 scores like the reference (hstu.py:965-979 + collector.py:245): bf16 scorer for the candidates
            # within 2^-7 of the k-th score, fp32 re-score of those, exact select (ops.catalog_topk_exact)
            if defer:
                # everything up to the verification read is enqueued (capturable); finish() reads the flags, repairs the rare rows
                vals, idx, fin = ops.catalog_topk_exact(heads_n.reshape(B * H, -1).float().contiguous(), H, items_bf, self._item_cache[3],
                                                        tag_bits, row_bits, hist_ptr, hist_items, k, n_items=all_item_feature.shape[0],
                                                        defer=True)

                def finish():
                    v, i = fin()
                    return FusedTopK(v.view(B, H, k), i.view(B, H, k), logs)
                return finish
            vals, idx = ops.catalog_topk_exact(heads_n.reshape(B * H, -1).float().contiguous(), H, items_bf, self._item_cache[3],
                                               tag_bits, row_bits, hist_ptr, hist_items, k, n_items=all_item_feature.shape[0],
                                               stats=stats)
        elif defer:
            raise ValueError("deferred decode needs exact_fp32_topk")
        else:
            vals, idx = ops.catalog_topk(users, H, items_bf, tag_bits, row_bits, hist_ptr, hist_items, k, stats=stats,
                                         n_items=all_item_feature.shape[0])
        return FusedTopK(vals.view(B, H, k), idx.view(B, H, k), logs)

    @torch.no_grad()
    def _dense_scores(self, last, all_item_feature, all_item_tags, target_tags, save_for_eval, heads_n=None):
        """Reference contract of `predict`: dense scores [B,H,N] fp32 with -inf masks (hstu.py:915-1016 / hllm.py:785-883)."""
        S, C = self.num_segment_head, self.num_prior_head
        pred = None
        if heads_n is not None:
            heads = heads_n
        elif self.prior_switch is not None:
            heads, pred = self._heads_at_last(last, want_switch=True)                              # [B,H,D] fp32, [B,n] bool
        else:
            heads = self._heads_at_last(last)
        feat = all_item_feature.float()
        feat = feat / feat.norm(dim=-1, keepdim=True)
        scores = torch.matmul(heads, feat.t())
        if self.loss == 'prior':
            additive = self.head_interaction == 'additive'
            if self.prior_given_at_test:
                given = target_tags[:, :self.given_prior_len].bool().any(dim=1)
                if additive:
                    scores[:, S:].masked_fill_(~given.unsqueeze(-1), float('-inf'))
                else:
                    scores.masked_fill_(~given.repeat(1, S).unsqueeze(-1), float('-inf'))
            tagm = all_item_tags.bool()
            if additive:
                scores[:, S:].masked_fill_(~tagm.unsqueeze(0), float('-inf'))
            else:
                scores.masked_fill_(~tagm.repeat(S, 1).unsqueeze(0), float('-inf'))
            if pred is not None and self.use_prior_switch_test:
                off = self._switched_off(pred)
                if additive:
                    scores[:, S:].masked_fill_(off.unsqueeze(-1), float('-inf'))
                else:
                    scores.masked_fill_(off.repeat(1, S).unsqueeze(-1), float('-inf'))
        wandb_logs = {'num_samples': self.eval_pred_len * heads.shape[0]}
        if pred is not None:
            self._switch_logs(pred, target_tags, wandb_logs)
        saved_user = saved_head = None
        if save_for_eval:
            saved_user = None if last is None else last.float().cpu().numpy()
            saved_head = heads.float().cpu().numpy()
        return scores, wandb_logs, saved_user, saved_head
