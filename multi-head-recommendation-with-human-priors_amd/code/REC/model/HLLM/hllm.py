"""HLLM with multi-head prior-guided decoding, MI355X-native: the HSTU path's twin on an LLM user decoder.

Same registry surface (`get_model("HLLM")`), constructor, `forward(interaction, mode=...)` / `predict` /
`compute_item` / `set_all_item_embeds` contract and head / `logit_scale` state_dict names as the reference
`code/REC/model/HLLM/hllm.py` (class HLLM, lines 40-908).  Heads, sampled-softmax loss, prior switch and the fused
catalog decode are the SAME code and kernels as HSTU (REC/model/multihead.py - the reference carries that code twice);
the sequence encoder is `user_llm(inputs_embeds=...)` (hllm.py:501-502, 781-783) on the native Llama decoder
(REC/model/HLLM/modeling_llama.py: RMSNorm / RoPE / causal softmax attention / SwiGLU in gfx950 kernels); item
embeddings come from the frozen tower's cache `all_item_embeds` (hllm.py:485-488, 528-531, 627-630) through the
embedding-gather kernel, or from the item tower itself (`forward_item_emb`, hllm.py:399-464) on packed
`cu_input_lens` token batches.

Scope notes: text-only Llama-architecture towers (`model_type` llama / mistral / qwen2 - e.g. TinyLlama-1.1B,
Qwen2.5-1.5B, the user towers of the reference's scripts), Baichuan2 (`model_type` baichuan: packed `W_pack` projection)
and the reference's `dummy_llm` debug tower; the Bert / vision variants of `create_llm` (hllm.py:325-376) raise NotImplementedError naming the architecture.  Pretrained weights are read from
`<dir>/model.safetensors` when present; there is no network here, so synthetic runs pass `user_llm_config` /
`item_llm_config` dicts instead of directories.  There is no CPU path.
"""
from logging import getLogger

import torch
import torch.nn as nn

from REC.model.basemodel import BaseModel, all_gather, all_gather_ids
from REC.model.HLLM.modeling_llama import LlamaConfig, LlamaForCausalLM
from REC.model.multihead import MultiHeadDecoding
from REC.utils.enum_type import InputType


class DummyLLM(nn.Module):
    """The reference's debug tower (REC/model/layers.py:94-115): embedding + one Linear."""

    def __init__(self, input_size, hidden_size):
        super().__init__()
        self.input_layer = nn.Embedding(input_size, hidden_size)
        self.embed_layer = nn.Linear(hidden_size, hidden_size)
        self.hidden_size = hidden_size

    def forward(self, input_ids=None, inputs_embeds=None, **kwargs):
        from types import SimpleNamespace
        if input_ids is not None:
            assert inputs_embeds is None, "Cannot pass both input_ids and inputs_embeds"
            out = self.input_layer(input_ids)
        elif inputs_embeds is not None:
            out = inputs_embeds
        else:
            raise ValueError('You must provide either input_ids or inputs_embeds')
        with torch.autocast(device_type=out.device.type, dtype=torch.bfloat16, enabled=out.is_cuda):
            out = self.embed_layer(out)
        return SimpleNamespace(hidden_states=[out])

    def get_input_embeddings(self):
        return self.input_layer


class HLLM(MultiHeadDecoding, BaseModel):
    input_type = InputType.SEQ

    def __init__(self, config, dataload):
        super().__init__()
        self.logger = getLogger()
        self.dummy_llm = bool(config['dummy_llm'])
        self.freeze_item_llm = config.get('freeze_item_llm', False)
        self.all_item_embeds = None
        self._item_table = None                      # fp32 contiguous copy of all_item_embeds for the gather kernel
        self.max_seq_length = config['MAX_ITEM_LIST_LENGTH']
        self.item_pretrain_dir = config['item_pretrain_dir']
        self.user_pretrain_dir = config['user_pretrain_dir']
        self.gradient_checkpointing = config['gradient_checkpointing']        # accepted, unused (see modeling_llama.py)
        self.use_ft_flash_attn = config['use_ft_flash_attn']                  # accepted: the fused attention is the only path
        self.item_llm = self.create_llm(self.item_pretrain_dir, config['item_llm_init'], config['item_llm_config'],
                                        optional=bool(self.freeze_item_llm))
        self.user_llm = self.create_llm(self.user_pretrain_dir, config['user_llm_init'], config['user_llm_config'])
        self.item_emb_token_n = config['item_emb_token_n'] if config['item_emb_token_n'] is not None else 1
        if self.item_emb_token_n > 1:
            raise NotImplementedError(f"Not support item_emb_token_n {self.item_emb_token_n} > 1")
        self.user_llm_hidden_size = self._hidden_size(self.user_llm)
        self.item_llm_hidden_size = self._hidden_size(self.item_llm) if self.item_llm is not None else self.user_llm_hidden_size
        if self.item_emb_token_n > 0 and self.item_llm is not None:
            self.item_emb_tokens = nn.Parameter(torch.zeros(1, self.item_emb_token_n, self.item_llm_hidden_size))
            self.item_emb_tokens.data.normal_(mean=0.0, std=0.02)
        else:                                        # mean pooling (or no item tower in this process)
            self.item_emb_tokens = None

        # heads, loss configuration, prior switch: shared with HSTU (REC/model/multihead.py); hllm.py:120-122: ln(1/0.07)
        self._init_multihead(config, dataload, self.user_llm_hidden_size, temp_init=1 / 0.07)
        if self.freeze_item_llm and self.item_llm is not None:
            for p in self.item_llm.parameters():
                p.requires_grad_(False)
            if self.item_emb_tokens is not None:
                self.item_emb_tokens.requires_grad_(False)
        self.sparse_grad = None                      # no trainable item table: the fused optimizer's sparse path is idle

    @staticmethod
    def _hidden_size(llm):
        return llm.hidden_size if isinstance(llm, DummyLLM) else llm.config.hidden_size

    def create_llm(self, pretrain_dir, init=True, cfg_dict=None, optional=False):
        """hllm.py:294-376.  `cfg_dict` (config keys `user_llm_config` / `item_llm_config`) stands in for
        `<pretrain_dir>/config.json` when no directory exists (synthetic / random-init runs)."""
        if cfg_dict is None and pretrain_dir is None:
            if optional:
                return None
            raise ValueError("either a pretrain dir with config.json or an *_llm_config dict is required")
        lcfg = LlamaConfig(**cfg_dict) if cfg_dict is not None else LlamaConfig.from_pretrained(pretrain_dir)
        if lcfg.model_type == "baichuan":                            # hllm.py:333-340
            from REC.model.HLLM.baichuan.configuration_baichuan import BaichuanConfig
            lcfg = BaichuanConfig(**cfg_dict) if cfg_dict is not None else BaichuanConfig.from_pretrained(pretrain_dir)
        if self.dummy_llm:
            self.logger.info('Using a dummy LLM for debugging...')
            return DummyLLM(lcfg.vocab_size, lcfg.hidden_size)
        if lcfg.model_type not in ("llama", "qwen2", "mistral", "baichuan"):   # same decoder maths (qwen2: + q/k/v biases,
            raise NotImplementedError(f"create_llm: architecture '{lcfg.model_type}' is not built on the MI355X path yet "
                                      "(llama / qwen2 / mistral / baichuan text decoders only)")          # baichuan: packed q|k|v)
        cls = LlamaForCausalLM
        if lcfg.model_type == "baichuan":
            from REC.model.HLLM.baichuan.modeling_baichuan import BaichuanForCausalLM as cls
        if init and cfg_dict is None:
            return cls.from_pretrained(pretrain_dir, config=lcfg)
        return cls(lcfg)

    # ------------------------------------------------------------------------------------------
    # item tower
    # ------------------------------------------------------------------------------------------
    def forward_item_emb(self, input_ids, position_ids, cu_input_lens, emb_token_n, emb_tokens, llm,
                         pixel_values=None, image_grid_thw=None):
        """Packed token batch of n items -> one embedding per item [n, D] (hllm.py:399-464): the hidden state at the
        item's last token (the learnable [ITEM] token) or the mean over its tokens."""
        if pixel_values is not None and pixel_values.numel() > 0:
            raise NotImplementedError("image inputs (llava / qwen2-vl towers) are not built on the MI355X path")
        emb_pos = cu_input_lens.cumsum(dim=0, dtype=torch.int32)
        out = llm(input_ids=input_ids, position_ids=position_ids.unsqueeze(0), cu_input_lens=cu_input_lens,
                  emb_tokens=emb_tokens, emb_pos=emb_pos, emb_token_n=emb_token_n)
        hidden = out.hidden_states[-1].squeeze(0)
        if self.dummy_llm:
            return hidden[:cu_input_lens.size(0)]
        if emb_token_n > 0:
            return hidden[(emb_pos - 1).long()]
        seg = torch.repeat_interleave(torch.arange(cu_input_lens.numel(), device=hidden.device), cu_input_lens.long())
        summed = torch.zeros(cu_input_lens.numel(), hidden.shape[-1], dtype=torch.float32, device=hidden.device)
        summed.index_add_(0, seg, hidden.float())
        return summed / cu_input_lens.unsqueeze(1)

    def _item_embs(self, interaction, prefix, n_rows, suffix=''):
        """[n_rows, D] fp32 embeddings of the items under the `<prefix>_*<suffix>` keys (hllm.py:491-498, 534-543, 633-643),
        through the item tower."""
        k = lambda name: interaction.get(f'{prefix}_{name}{suffix}')        # noqa: E731
        e = self.forward_item_emb(k('input_ids'), k('position_ids'), k('cu_input_lens'), self.item_emb_token_n,
                                  self.item_emb_tokens, self.item_llm, k('pixel_values'), k('image_grid_thw'))
        return e.float().reshape(n_rows, -1)

    @torch.no_grad()
    def set_all_item_embeds(self, all_item_embeds):
        self.all_item_embeds = all_item_embeds.detach()
        self._item_table = None
        self._item_cache = None

    def _table(self):
        if self.all_item_embeds is None:
            raise RuntimeError("freeze_item_llm: call set_all_item_embeds(<[N, D] item embeddings>) first (hllm.py:889-891)")
        if self._item_table is None:
            self._item_table = self.all_item_embeds.float().contiguous()
        return self._item_table

    def train(self, mode=True):
        self._item_cache = None
        return super().train(mode)

    @torch.no_grad()
    def compute_item_all(self):
        self._item_cache = None
        return self.all_item_embeds

    @torch.no_grad()
    def compute_item(self, interaction):
        """hllm.py:893-908: embeddings of one batch of items for the whole-catalog pass (trainer.py:735-786)."""
        n = interaction['pos_cu_input_lens'].size(0)
        return (self._item_embs(interaction, 'pos', n), interaction['pos_tag_categories'],
                interaction['pos_original_tag_categories'])

    # ------------------------------------------------------------------------------------------
    # training
    # ------------------------------------------------------------------------------------------
    def forward(self, interaction, mode='train'):
        if mode == 'predict':
            return self.predict(*interaction)
        if mode == 'compute_item':
            return self.compute_item(interaction)
        from mhr_amd import ops
        from REC.model.hstu_functional import L2NormFn
        mask = interaction['attention_mask'].bool()
        if not mask.is_cuda:
            raise RuntimeError("HLLM.forward runs on the MI355X only (no CPU path); move the batch to the GPU")
        B, S = mask.shape
        L, P, D, C = self.max_seq_length, self.pred_len, self._head_dim, self.num_prior_head
        assert S == L + P, f"attention_mask is [B, {S}], expected MAX_ITEM_LIST_LENGTH + pred_len = {L + P}"
        self._clamp_logit_scale()
        pools = self._negative_pools(C + 1)          # pool C = the shared pool `neg_item_ids`, pool c = `neg_item_ids_cat{c}`
        if self.freeze_item_llm:
            # hllm.py:485-488, 528-531, 627-630: lookups in the frozen tower's cache; negative IDS are shared across ranks
            ids_of = {p: interaction['neg_item_ids' if p == C else f'neg_item_ids_cat{p}'] for p in pools}
            pool_ids = [all_gather_ids(ids_of[p].contiguous()).reshape(-1) for p in pools]
            ids_all = torch.cat([interaction['pos_item_ids'].reshape(-1)] + pool_ids).contiguous()
            rows_all, _ = ops.embedding_gather(self._table(), ids_all, torch.float32)
            e_rows = rows_all[:B * S]
            n_pool = pool_ids[0].numel()
            negs_pools = ops.l2norm_rows(rows_all[B * S:].contiguous(), torch.bfloat16).view(len(pools), n_pool, D)
        else:
            # trainable item tower: embeddings (and their gradients) come from the tower; the negatives of every rank
            # are exchanged WITH gradients, as in the reference (hllm.py:545-547, 645-646)
            e_rows = self._item_embs(interaction, 'pos', B * S)
            negs = []
            for p in pools:
                sfx = '' if p == C else f'_cat{p}'
                n = self._item_embs(interaction, 'neg', interaction[f'neg_cu_input_lens{sfx}'].numel(), sfx)
                negs.append(all_gather(n, sync_grads=True).reshape(-1, D))
            negs_pools = L2NormFn.apply(torch.stack(negs).reshape(-1, D).contiguous()).view(len(pools), -1, D)
        x = e_rows.view(B, S, D)[:, :L]
        out = self.user_llm(inputs_embeds=x, attention_mask=mask[:, :L]).hidden_states[-1].float()     # hllm.py:501-502
        return self._multihead_loss(out, e_rows, negs_pools, pools, mask, interaction.get('pos_tag_categories'))

    # ------------------------------------------------------------------------------------------
    # evaluation
    # ------------------------------------------------------------------------------------------
    @torch.no_grad()
    def _last_hidden(self, item_seq, all_item_feature):
        """hllm.py:771-783: history embeddings looked up in the item table, user decoder, last position -> [B,D] fp32."""
        from mhr_amd import ops
        if all_item_feature.dtype == torch.float32 and all_item_feature.is_contiguous():
            emb, _ = ops.embedding_gather(all_item_feature, item_seq.contiguous(), torch.float32)
        else:
            emb = all_item_feature[item_seq].float()
        out = self.user_llm(inputs_embeds=emb, attention_mask=(item_seq > 0)).hidden_states[-1]
        return out[:, -1].float()

    @torch.no_grad()
    def predict_topk(self, item_seq, all_item_feature, all_item_tags, target_tags, history=None, k=200,
                     suppress_history=True, stats=None):
        """Fused eval (see MultiHeadDecoding._decode_topk): replaces hllm.py:785-883 + trainer.py:724-726 + collector.py:245."""
        return self._decode_topk(self._last_hidden(item_seq, all_item_feature), all_item_feature, all_item_tags, target_tags,
                                 history, k, suppress_history, stats, all_item_feature.shape[0])

    @torch.no_grad()
    def predict(self, item_seq, time_seq, all_item_feature, all_item_tags, target_tags, save_for_eval=False):
        """Reference contract: dense scores [B,H,N] fp32 with -inf masks (hllm.py:765-883)."""
        return self._dense_scores(self._last_hidden(item_seq, all_item_feature), all_item_feature, all_item_tags,
                                  target_tags, save_for_eval)
