"""HSTU with multi-head prior-guided decoding, MI355X-native.

Same registry surface, constructor, `forward` / `predict` / `compute_item_all` contract and `state_dict`
names as the reference `code/REC/model/IDNet/hstu.py` (class HSTU, lines 331-1030), so `get_model("HSTU")`,
the reference YAML presets and checkpoints keep working.  The arithmetic is not torch eager: embedding
gather / sparse gradient, LayerNorm, the pointwise-gated attention, the gate, the sampled-softmax loss and
the catalog scoring + top-k run in hand-written gfx950 kernels (mhr_amd.ops); only the dense projections are
library GEMMs.  There is no CPU path: on a machine without the HIP library construction still works (for
state_dict handling) but forward/predict raise.

Differences from the reference that are deliberate:
  * the relative position/time bias parameters are kept for checkpoint compatibility and never applied
    (the reference builds them and never calls them, hstu.py:187-189 vs 221-290);
  * `forward` never synchronises with the host: token sets stay at fixed capacity with device-side counts
    (the reference's `mask.sum() == 0` / `.any()` branches, hstu.py:722, 815, 861, each cost a sync);
  * the item table's gradient is left as a deterministic sparse row set on `self.sparse_grad`
    (set `dense_embedding_grad = True` for the reference's dense `weight.grad`);
  * `predict_topk` returns per-head top-k without materialising the [B,H,N] score tensor; `predict`
    still returns dense scores for callers written against the reference.
"""
import os
from logging import getLogger

import torch
import torch.nn as nn
import torch.nn.functional as F

from REC.model.basemodel import BaseModel, all_gather_pool_ids
from REC.model.multihead import FusedTopK, MultiHeadDecoding  # noqa: F401  (FusedTopK re-exported)
from REC.utils.enum_type import InputType

EARLY_LOSS_PREP = os.environ.get("MHR_EARLY_LOSS_PREP", "1") != "0"       # 0: the loss preparation runs in line, after the encoder
FUSED_NEG_GATHER = os.environ.get("MHR_FUSED_NEG_GATHER", "1") != "0"     # 0: gather the negative pools as fp32 rows, normalise after
WEIGHT_GRAD_STACK = os.environ.get("MHR_WEIGHT_GRAD_STACK", "1") != "0"   # 0: every layer's weight gradients inside its own backward


def truncated_normal(x, mean, std):
    """Four normal draws per element, the first inside (-2, 2) wins (reference hstu.py:23-31)."""
    with torch.no_grad():
        draws = torch.empty(x.shape + (4,), dtype=x.dtype, device=x.device).normal_()
        inside = (draws < 2) & (draws > -2)
        first = inside.to(torch.uint8).argmax(dim=-1, keepdim=True)
        x.copy_(draws.gather(-1, first).squeeze(-1).mul_(std).add_(mean))
    return x


class RelativeBucketedTimeAndPositionBasedBias(nn.Module):
    """Parameter holder only: `_ts_w` [num_buckets+1], `_pos_w` [2*max_seq_len-1] (reference hstu.py:74-131).
    The reference never applies this bias in its attention, so neither do we."""

    def __init__(self, max_seq_len, num_buckets):
        super().__init__()
        self._ts_w = nn.Parameter(torch.empty(num_buckets + 1).normal_(mean=0, std=0.02))
        self._pos_w = nn.Parameter(torch.empty(2 * max_seq_len - 1).normal_(mean=0, std=0.02))


class SequentialTransductionUnitJagged(nn.Module):
    """One HSTU block's parameters: `_uvqk` [D, 4D], `_o` Linear(D, D) (reference hstu.py:163-211)."""

    def __init__(self, embedding_dim, n_heads, dropout_ratio, rel_bias=None, eps=1e-6):
        super().__init__()
        self._embedding_dim = embedding_dim
        self._num_heads = n_heads
        self._dropout_ratio = dropout_ratio
        self._eps = eps
        self._rel_attn_bias = rel_bias
        self._uvqk = nn.Parameter(torch.empty(embedding_dim, 4 * embedding_dim).normal_(mean=0, std=0.02))
        self._o = nn.Linear(embedding_dim, embedding_dim)
        nn.init.xavier_uniform_(self._o.weight)


class HSTUJagged(nn.Module):
    def __init__(self, modules):
        super().__init__()
        self._attention_layers = nn.ModuleList(modules)


class HSTU(MultiHeadDecoding, BaseModel):
    input_type = InputType.SEQ

    @property
    def graph_capable(self):
        """The training step can be replayed from a hipGraph: the streaming path (feature dims up to 256) never synchronises
        with the host and reads its dropout step counter from `_seed_dev`; the wide path bounds a chunk loop on the host."""
        from mhr_amd import ops
        return type(self).__name__ == "HSTU" and self._hstu_embedding_dim in ops.STREAM_DIMS

    def __init__(self, config, dataload):
        super().__init__()
        self.logger = getLogger()
        self.item_num = dataload.item_num
        self._item_embedding_dim = config['item_embedding_size']
        self._hstu_embedding_dim = D = config['hstu_embedding_size']
        self.max_seq_length = L = config['MAX_ITEM_LIST_LENGTH']
        self._num_blocks = config['n_layers']
        self._num_heads = config['n_heads']
        self._dqk = self._dv = D // config['n_heads']
        self._linear_activation = config['hidden_act'] if config['hidden_act'] else 'silu'
        if self._linear_activation != 'silu':
            raise NotImplementedError("only hidden_act='silu' is implemented in the fused kernels")
        self._linear_dropout_rate = config['hidden_dropout_prob'] or 0.0
        self._attn_dropout_rate = config['attn_dropout_prob']          # stored, unused (as in the reference)
        self._enable_relative_attention_bias = bool(config['enable_relative_attention_bias'])

        self.position_embedding = nn.Embedding(L + 1, D)
        self._hstu = HSTUJagged([
            SequentialTransductionUnitJagged(
                D, self._num_heads, self._linear_dropout_rate,
                RelativeBucketedTimeAndPositionBasedBias(2 * L, 128) if self._enable_relative_attention_bias else None)
            for _ in range(self._num_blocks)])
        self.item_embedding = nn.Embedding(self.item_num, self._item_embedding_dim, padding_idx=0)
        self.item_id_proj_tower = (nn.Identity() if self._item_embedding_dim == D
                                   else nn.Linear(self._item_embedding_dim, D, bias=False))

        # heads, loss configuration, prior switch: shared with the HLLM twin (REC/model/multihead.py)
        self._init_multihead(config, dataload, D, temp_init=1 / 0.05)

        self.register_buffer("_attn_mask", torch.triu(torch.ones((L, L), dtype=torch.bool), diagonal=1))
        self._verbose = False
        self.reset_params()

        # runtime state of the fused path
        self.dense_embedding_grad = False
        self.sparse_grad = None
        self._row_slot = None
        self._side_stream = None
        self._presorted = None
        self._pending_rows = None
        self._shared_pending = None
        self._row_exchange = None
        self._dp_ids = None
        self._bf16_cache = {}
        self._step_seed = 0
        self._seed_dev = None              # device int64[1] while the Trainer captures the step as a hipGraph

    # ------------------------------------------------------------------------------------------
    def reset_params(self):
        """trunc-normal(0.02) for every parameter outside the encoder (reference hstu.py:574-588)."""
        for name, p in self.named_parameters():
            if ("_hstu" in name) or ("_embedding_module" in name) or ('logit_scale' in name):
                continue
            truncated_normal(p.data, mean=0.0, std=0.02)

    def get_attention_mask(self, item_seq, bidirectional=False):
        """Kept for callers of the reference API; the fused attention takes the key-valid vector instead."""
        m = (item_seq != 0).unsqueeze(1).unsqueeze(2)
        if not bidirectional:
            m = torch.tril(m.expand((-1, -1, item_seq.size(-1), -1)))
        return m

    # ------------------------------------------------------------------------------------------
    # encoder + heads
    # ------------------------------------------------------------------------------------------
    def _layer_weights_bf16(self, i, layer):
        """bf16 operands of layer i's two GEMMs.  While the module is in eval mode the casts are done once and kept
        (weights do not move between `eval()` and the next `train()` / `load_state_dict`, which drop the cache)."""
        if self.training or torch.is_grad_enabled():
            return None                                   # SplitKLinearFn casts (and differentiates through) the masters
        hit = self._bf16_cache.get(i)
        if hit is None:
            hit = self._bf16_cache[i] = (layer._uvqk.to(torch.bfloat16), layer._o.weight.to(torch.bfloat16),
                                         layer._o.bias.to(torch.bfloat16))
        return hit

    def _table_catch_up(self, ids):
        opt = getattr(self, "_table_optimizer", None)
        if opt is not None:
            opt.catch_up(ids)

    def sync_table(self):
        """With the lazy table optimizer (mhr_amd.optim.FusedAdamW(lazy_table=True)) rows lag behind until they are read by a
        training forward; this brings the whole table up to date.  Called on eval(), state_dict() and by the Trainer."""
        opt = getattr(self, "_table_optimizer", None)
        if opt is not None:
            opt.flush_table()

    def state_dict(self, *args, **kwargs):
        self.sync_table()
        return super().state_dict(*args, **kwargs)

    def train(self, mode=True):
        if not mode:
            self.sync_table()
        self._bf16_cache = {}
        self._item_cache = None            # keyed on the feature tensor's address: a new table may reuse a freed one's
        return super().train(mode)

    def load_state_dict(self, *args, **kwargs):
        self._bf16_cache = {}
        self._item_cache = None
        out = super().load_state_dict(*args, **kwargs)
        opt = getattr(self, "_table_optimizer", None)
        if opt is not None:                              # loaded rows are current: nothing to replay
            opt.last_step.fill_(opt.step_count)
        return out

    def _dead_rows_allowed(self):
        """May the encoder leave the rows in front of a sequence's first valid key unread?  Yes while nothing reads the hidden
        states of padding positions: the sampled-softmax tokens, the decode and the last-position switch are all on valid rows.
        The prior switch's category loss over ALL positions (reference hstu.py:757-806: a BCE over [batch, seq_len] without a
        mask, unless `switch_last_only`) does read them - and sends gradient into them - so those models keep every row."""
        return getattr(self, "prior_switch", None) is None or bool(getattr(self, "switch_last_only", False))

    def _encode(self, x, key_valid, training=None, want_bf16=False):
        """x [B,L,D] fp32, key_valid [B,L] uint8 -> [B,L,D] fp32 (reference hstu.py:221-328); want_bf16: (out, its bf16 copy -
        the operand of the decoding heads' GEMM, written by the last residual add's own pass)."""
        from mhr_amd import ops
        from REC.model.hstu_functional import (AddCastFn, AddLayerNormFn, HSTUCoreFn, LayerNormFn, LayerNormResidualFn, SplitKLinearFn,
                                               WeightGradStack, ROWS_GEMM, _rows_gemm_pays)
        B, L, D = x.shape
        x2 = x.reshape(B * L, D)
        training = self.training if training is None else training
        p = self._linear_dropout_rate if training else 0.0
        layers = self._hstu._attention_layers
        n = len(layers)
        if ops.SEQ_LAYOUT and n and getattr(key_valid, "_mhr_layout", None) is None:
            key_valid._mhr_layout = ops.attn_seq_layout(key_valid, B, L)     # front padding: the layers skip the dead blocks
        lay = getattr(key_valid, "_mhr_layout", None)
        key_valid._mhr_dead_ok = self._dead_rows_allowed()
        # rows in front of a sequence's first valid key: the layers' row-wise kernels do not load them (zeros in, zeros out)
        dead = (lay[2], L) if (ops.DEAD_ROWS and lay is not None and len(lay) > 2 and self._dead_rows_allowed()) else None
        # PACKED rows: the batch carries a static capacity for its valid positions (`_mhr_rows_cap` on the mask: a bucketed count
        # the loader knows on the host) - the layers then run over the valid rows only, back to back in a [capacity, D] buffer
        # (csrc/rows_pack.hip), the attention addresses the sequences through cu_rows.  Same condition as the dead rows: nothing
        # may read the hidden states of padding positions.  Data parallel: the capacity is part of the batch signature the step
        # graphs are keyed on, and every rank must capture and replay in the same steps (a capture issues no collective) - the
        # loaders must hand all ranks the SAME capacity for a step (the maximum over the ranks; bench.py does that).
        cap = getattr(key_valid, "_mhr_rows_cap", None)
        pack = None
        if (cap is not None and ops.PACK_ROWS and n and lay is not None and self._dead_rows_allowed() and 0 < int(cap) < B * L
                and D % 8 == 0 and self._packable(L)):
            from REC.model.hstu_functional import MoveRowsFn
            cap = int(cap)
            cu, src_of, row_of, overflow = ops.seq_pack_maps(key_valid, B, L, cap)
            # a capacity below the batch's valid positions would silently drop rows: the first host-issued steps at a capacity
            # read the count back (a captured step cannot, and later steps trust the loader)
            seen = self.__dict__.setdefault("_pack_checked", {})
            if not torch.cuda.is_current_stream_capturing() and seen.get(cap, 0) < 4:
                seen[cap] = seen.get(cap, 0) + 1
                if int(overflow.item()) != 0:
                    raise RuntimeError(f"packed encoder rows: the batch has {int(overflow.item())} valid positions, "
                                       f"its capacity hint (_mhr_rows_cap) says {cap}")
            pack = (src_of, row_of)
            x2 = MoveRowsFn.apply(x2, src_of, row_of)                      # [cap, D]
            kv_w = key_valid
            key_valid = self._ones_u8(cap, x.device)
            key_valid._mhr_layout = (None, lay[1], None, cu)                # sequences longest first, addressed through cu_rows
            key_valid._mhr_dead_ok = False
            dead = None
        R = x2.shape[0]
        # training with the fused optimizer in its one-backward-per-step mode: the layers' weight-gradient products are formed
        # for all layers at once at the step (WeightGradStack); the kernels below write their operands straight into its slices
        stack = None
        opt = getattr(layers[0]._uvqk, "_mhr_opt", None) if n else None
        if (WEIGHT_GRAD_STACK and D <= 512 and opt is not None and torch.is_grad_enabled()       # (wider layers: their own products fill the chip)
                and self._layer_weights_bf16(0, layers[0]) is None
                and all(getattr(q, "_mhr_direct_grad", False) and q.grad is not None and getattr(q, "_mhr_opt", None) is opt
                        for ly in layers for q in (ly._uvqk, ly._o.weight))
                and all(tuple(ly._uvqk.shape) == (D, 4 * D) and tuple(ly._o.weight.shape) == (D, D) for ly in layers)):
            stack = WeightGradStack(opt, [ly._uvqk for ly in layers], [ly._o.weight for ly in layers], R, D, x.device)
        sl = (lambda buf, i: buf[i]) if stack is not None else (lambda buf, i: None)
        if n > 1 and x2.requires_grad and x2.is_contiguous():
            x2, xn = LayerNormResidualFn.apply(x2, layers[0]._eps, True, sl(stack and stack.xn, 0))
        else:
            xn = LayerNormFn.apply(x2, layers[0]._eps, sl(stack and stack.xn, 0))
        for i, layer in enumerate(layers):
            cached = self._layer_weights_bf16(i, layer)
            # dropout seed = (step * 1000003 + layer part) & (2^63 - 1); a hipGraph-replayed step reads the step counter from
            # device memory (self._seed_dev, set by the Trainer's step graph) and passes only the layer part
            seed_dev = getattr(self, "_seed_dev", None) if training else None
            layer_part = i * 7919 + self.rank * 104729
            seed = layer_part if seed_dev is not None else (self._step_seed * 1000003 + layer_part) & 0x7FFFFFFFFFFFFFFF
            if cached is None:
                h = SplitKLinearFn.apply(xn, layer._uvqk, None, False, None, (stack, "uvqk", i) if stack is not None else None)
                o = HSTUCoreFn.apply(h, key_valid, B, L, self._num_heads, self._dqk, layer._eps, p, seed, seed_dev,
                                     sl(stack and stack.o, i), sl(stack and stack.dh, i))
                y = SplitKLinearFn.apply(o, layer._o.weight, layer._o.bias, True, None, (stack, "o", i) if stack is not None else None)
            else:
                h = ops.rows_gemm(xn, cached[0], None, w_is_kn=True) if (ROWS_GEMM and _rows_gemm_pays(xn, cached[0])) else xn @ cached[0]
                o = HSTUCoreFn.apply(h, key_valid, B, L, self._num_heads, self._dqk, layer._eps, p, seed)
                y = F.linear(o, cached[1], cached[2])
            if i + 1 < n:                      # residual add + the next layer's LayerNorm in one pass
                x2, xn = AddLayerNormFn.apply(x2, y, layers[i + 1]._eps, sl(stack and stack.xn, i + 1), sl(stack and stack.dy, i), dead)
            elif want_bf16 and y.dtype == torch.bfloat16 and x2.numel() % 8 == 0:
                x2, x16 = AddCastFn.apply(x2, y, sl(stack and stack.dy, i))
                if pack is not None:                        # back to the windows: padding positions read as zeros
                    x2, x16 = MoveRowsFn.apply(x2, pack[1], pack[0]), MoveRowsFn.apply(x16, pack[1], pack[0])
                return x2.view(B, L, D), x16.view(B, L, D)
            else:
                x2 = torch.add(x2, y)           # fp32 + bf16 -> fp32 in one kernel
        if pack is not None:
            x2 = MoveRowsFn.apply(x2, pack[1], pack[0])
        return (x2.view(B, L, D), None) if want_bf16 else x2.view(B, L, D)

    def _packable(self, L):
        """Packed rows need the attention's resident form (sequence offsets are not wired into the streamed one)."""
        hd = self._dqk
        nks = (hd + 15) // 16
        nb = (L + 31) // 32
        return 2 * nb * 32 * nks * 32 + nb * 4 + 16 <= 64 * 1024 and 4 * nb * 32 * nks * 32 + nb * 4 + 16 + 4 * 32 * 36 * 4 <= 80 * 1024 and nks <= 4

    def _ones_u8(self, n, device):
        buf = getattr(self, "_ones_buf", None)
        if buf is None or buf.numel() < n or buf.device != device:
            buf = self._ones_buf = torch.ones(max(n, 1), dtype=torch.uint8, device=device)
        return buf[:n]

    # ------------------------------------------------------------------------------------------
    # training
    # ------------------------------------------------------------------------------------------
    def forward(self, interaction):
        from REC.model.hstu_functional import EmbeddingGatherFn, GatherL2NormFn, L2NormFn
        items, neg_items, user_mask, pos_tags = interaction
        if not items.is_cuda:
            raise RuntimeError("HSTU.forward runs on the MI355X only (no CPU path); move the batch to the GPU")
        B = items.shape[0]
        L, P, D = self.max_seq_length, self.pred_len, self._hstu_embedding_dim
        mask = user_mask.bool()
        self._step_seed += 1
        early_prep = EARLY_LOSS_PREP and torch.is_grad_enabled()
        if not early_prep:
            self._clamp_logit_scale()

        # which negative pools the loss reads (reference hstu.py:669-670, 751-752); ids are shared across ranks
        pools = self._negative_pools(neg_items.shape[1])
        from mhr_amd import distributed as dist_
        # data parallel: this rank's item-window ids ride the id all-gather of the negative pools, so every rank knows the ids of
        # ALL ranks' gradient rows now - the id sort of the sparse-row reduction runs early (below), the backward exchanges rows only
        dp_ids = (dist_.active() and torch.is_grad_enabled() and not getattr(self, "accumulate_rows", False)
                  and not self.dense_embedding_grad)
        if dp_ids:
            pool_ids, items_all = all_gather_pool_ids(neg_items, pools, extra=items.reshape(-1))
            self._dp_ids = torch.cat([items_all.reshape(-1)] + pool_ids).contiguous()       # the exchange's layout: private ids in rank order, then the shared ids
        else:
            pool_ids = all_gather_pool_ids(neg_items, pools)
            self._dp_ids = None
        n_pool = pool_ids[0].numel()
        n_item_ids = B * (L + P)
        ids_all = torch.cat([items.reshape(-1)] + pool_ids).contiguous()
        self._table_catch_up(ids_all)                      # lazy table optimizer: the rows this step reads, up to date first
        fused_pos = isinstance(self.item_id_proj_tower, nn.Identity)
        # the negative pools: gathered and L2-normalised in one pass (bf16 out) when the table rows are what the loss sees
        fused_negs = fused_pos and FUSED_NEG_GATHER and torch.is_grad_enabled() and not self.dense_embedding_grad
        rows_items, rows_negs, x = EmbeddingGatherFn.apply(self.item_embedding.weight,
                                              self.position_embedding.weight if fused_pos else None, ids_all,
                                              n_item_ids, L, L + P, self, fused_negs)
        if not fused_pos:
            rows_all = self.item_id_proj_tower(torch.cat([rows_items, rows_negs]))
            rows_items, rows_negs = rows_all[:n_item_ids], rows_all[n_item_ids:]
            x = rows_items.view(B, L + P, D)[:, :L] + self.position_embedding.weight[:L][None]
        e_rows = rows_items                                   # targets, [B*(L+P), D] fp32
        # data parallel: the negatives' gradient rows leave for their all-reduce from inside this backward (fused_pos:
        # they are rows of the table itself), underneath the encoder backward
        if fused_negs:
            negs_pools = GatherL2NormFn.apply(rows_negs, self.item_embedding.weight.detach(), ids_all[n_item_ids:], self)
        else:
            negs_pools = L2NormFn.apply(rows_negs.contiguous(), self if fused_pos else None)
        negs_pools = negs_pools.view(len(pools), n_pool, D)

        key_valid = mask[:, :L].to(torch.uint8).contiguous()
        if getattr(user_mask, "_mhr_rows_cap", None) is not None:  # the loader's static bound on the batch's valid positions (packed rows)
            key_valid._mhr_rows_cap = user_mask._mhr_rows_cap
        # everything of the loss that waits for nothing the encoder makes (token lists, row maps, the false-negative bit table of
        # the target rows, the accumulators of the loss backward, the id sort of the embedding backward) runs UNDERNEATH the
        # encoder on a second stream: some fifty few-microsecond launches and one 0.2 ms MFMA kernel on 104 workgroups that
        # otherwise sit on the step's critical path one after the other
        st, stages, side = {}, [], None
        capturing = torch.cuda.is_current_stream_capturing()
        segs = getattr(self, "_graph_segments", None) if capturing else None
        if early_prep:
            stages = self._early_loss_stages(st, pools, mask, pos_tags, e_rows, negs_pools, n_q_rows=B * self.medusa_num_heads * L)
            stages.insert(0, self._clamp_logit_scale)            # (read by the loss kernels only: behind the join)
            stages.append(lambda: self._presort_ids(ids_all if self._dp_ids is None else self._dp_ids))

        def run_stages():
            with torch.no_grad():
                while stages:
                    stages.pop(0)()

        if segs is not None and stages:
            # under the Trainer's capture: the stages become a hipGraph of their own on the capture's side stream, replayed
            # concurrently with the segment that holds the encoder (REC/trainer/trainer.py:_GraphSegments - every launch stays a
            # single-stream graph, which ROCm launches with one cheap host call; a graph with a forked branch cost 2.4 ms of
            # host time per replay, measured)
            segs.side_branch(run_stages)
            out, out16 = self._encode(x, key_valid, want_bf16=True)
            segs.join()
        elif stages and not capturing:
            # host-issued steps: a second stream forked here and joined in front of the heads; the encoder's launches go first
            # (the host is the slower side of the forward), the stages behind them
            if self._side_stream is None:
                self._side_stream = torch.cuda.Stream(device=items.device)
            side = self._side_stream
            side.wait_stream(torch.cuda.current_stream())        # the fork: everything up to here (rows, negatives) is visible
            out, out16 = self._encode(x, key_valid, want_bf16=True)
            with torch.cuda.stream(side):
                run_stages()
            torch.cuda.current_stream().wait_stream(side)
        else:                                                    # (somebody else's capture: one stream, the stages in line)
            run_stages()
            out, out16 = self._encode(x, key_valid, want_bf16=True)
        plan, early = st.get("plan"), st.get("early")
        return self._multihead_loss(out, e_rows, negs_pools, pools, mask, pos_tags, plan=plan, early=early, out_bf16=out16)

    def _presort_ids(self, ids_all):
        """The id sort and the zeroed row buffer of the sparse embedding backward (EmbeddingGatherFn.backward) depend on the
        batch's ids only: made early, next to the loss preparation (one-rank, one-backward-per-step training only - data
        parallel / accumulating steps sort the exchanged ids of all micro-batches at the optimizer step)."""
        from mhr_amd import distributed as dist_
        self._presorted = None
        if getattr(self, "accumulate_rows", False) or getattr(self, "dense_embedding_grad", False):
            return
        if dist_.active() and ids_all is not self._dp_ids:       # data parallel: only the exchanged id list (all ranks' ids) can be pre-sorted
            return
        sorted_ids, perm = torch.sort(ids_all)
        rows = torch.zeros(ids_all.numel(), self.item_embedding.weight.shape[1], dtype=torch.float32, device=ids_all.device)
        self._presorted = (ids_all, sorted_ids, perm, rows)

    def reset_step_state(self):
        """Forget what a half-finished step left for its backward / optimizer step (the Trainer calls this when a hipGraph
        capture raised partway through a step: the queued tensors belong to a capture that never ran)."""
        self._presorted = None
        self._pending_rows = None
        self._shared_pending = None
        self._row_exchange = None
        self._dp_ids = None
        self.sparse_grad = None

    def begin_sparse_exchange(self):
        """Data parallel: put the cross-rank exchange of the last backward's gradient rows on the wire (asynchronous;
        `finish_sparse_grad` completes it).  Called by the fused optimizer before its flat Adam."""
        if self._pending_rows is not None and self._row_exchange is None:
            from REC.model.hstu_functional import begin_pending_rows
            begin_pending_rows(self)

    def finish_sparse_grad(self):
        """The item table's gradient of the last backward as a SparseRowGrad (after the cross-rank exchange when
        data parallel).  Called by the fused optimizer."""
        if self._pending_rows is not None:
            from REC.model.hstu_functional import reduce_pending_rows
            reduce_pending_rows(self)
        return self.sparse_grad

    # ------------------------------------------------------------------------------------------
    # evaluation
    # ------------------------------------------------------------------------------------------
    @torch.no_grad()
    def compute_item_all(self):
        """L2-normalised (projected) item table, fp32 [N, D] (reference hstu.py:1018-1021)."""
        self.sync_table()                                   # lazy table optimizer: a read outside the training forward
        from mhr_amd import ops
        self._item_cache = None
        w = self.item_id_proj_tower(self.item_embedding.weight)
        return ops.l2norm_rows(w.contiguous(), torch.float32)

    @torch.no_grad()
    def _last_hidden(self, item_seq):
        """[B,L] ids (front zero padded) -> encoder output at the last position [B,D] fp32 (reference hstu.py:879-913)."""
        self.sync_table()                                   # lazy table optimizer: a read outside the training forward
        from mhr_amd import ops
        B, L = item_seq.shape
        if isinstance(self.item_id_proj_tower, nn.Identity):
            _, x = ops.embedding_gather(self.item_embedding.weight, item_seq.contiguous(), torch.float32,
                                        self.position_embedding.weight, L, torch.float32, want_rows=False)
        else:
            rows, _ = ops.embedding_gather(self.item_embedding.weight, item_seq.contiguous(), torch.float32)
            x = self.item_id_proj_tower(rows) + self.position_embedding.weight[:L][None]
        key_valid = (item_seq != 0).to(torch.uint8).contiguous()
        if getattr(item_seq, "_mhr_rows_cap", None) is not None and not torch.cuda.is_current_stream_capturing():
            key_valid._mhr_rows_cap = item_seq._mhr_rows_cap       # the loader's row capacity: the encoder runs on packed rows
        out = self._encode(x, key_valid, training=False)   # dropout off, like .eval()
        return out[:, -1]

    @torch.no_grad()
    def _user_heads(self, item_seq, want_switch=False):
        """-> L2-normalised head embeddings [B,H,D] fp32 (and the prior-switch decisions) (reference hstu.py:879-966)."""
        return self._heads_at_last(self._last_hidden(item_seq), want_switch)

    @torch.no_grad()
    def predict_topk(self, item_seq, all_item_feature, all_item_tags, target_tags, history=None, k=200,
                     suppress_history=True, stats=None, defer=False):
        """Fused eval: encoder -> heads -> catalog scoring with tag / pad / history masks -> exact per-head top-k.
        Replaces reference hstu.py:965-1015 + trainer.py:724-726 + collector.py:245 without the [B,H,N] tensor.
        defer=True: enqueue everything up to the decode's one verification read and return finish() -> FusedTopK (what the
        Trainer's replayed evaluation step captures)."""
        return self._decode_topk(self._last_hidden(item_seq), all_item_feature, all_item_tags, target_tags, history, k,
                                 suppress_history, stats, self.item_num, defer=defer)

    @torch.no_grad()
    def predict(self, item_seq, time_seq, all_item_feature, all_item_tags, target_tags, save_for_eval=False):
        """Reference contract: dense scores [B,H,N] fp32 with -inf masks (hstu.py:874-1016).  Kept for callers
        written against the reference; the fused path is `predict_topk`."""
        return self._dense_scores(self._last_hidden(item_seq), all_item_feature, all_item_tags, target_tags, save_for_eval)
