"""Llama-style decoder for the HLLM towers, MI355X-native.

Same module tree and `state_dict` names as the reference `code/REC/model/HLLM/modeling_llama.py` (HF layout:
`model.embed_tokens`, `model.layers.{i}.self_attn.{q,k,v,o}_proj`, `.mlp.{gate,up,down}_proj`, `.input_layernorm`,
`.post_attention_layernorm`, `model.norm`, `lm_head`), same call contract for the two ways HLLM uses it:

  * user decoder   `llm(inputs_embeds=[B,L,D], attention_mask=[B,L]).hidden_states[-1]`          (hllm.py:501-502, 781-783)
  * item tower     `llm(input_ids=[T], position_ids=[1,T], cu_input_lens=[n], emb_tokens=..., emb_pos=..., emb_token_n=k)
                    .hidden_states[-1]`   one packed row of n concatenated token sequences    (hllm.py:410-443)

The forward is not torch eager: RMSNorm fused with the residual adds, rotary embedding, causal softmax attention with
grouped KV heads (padded batches and packed `cu_input_lens` batches alike) and the SwiGLU gate run in hand-written gfx950
kernels (REC/model/llm_functional.py -> mhr_amd.ops); q/k/v and gate/up are each ONE library GEMM on concatenated
weights.  Deliberate differences from the reference: only the last hidden state is returned (`hidden_states` is a
1-tuple; every caller takes `[-1]`), the `lm_head` logits the reference computes and discards (modeling_llama.py:1254-
1255) are not computed, KV caches / generation are not part of the path, `gradient_checkpointing` is accepted and
ignored (288 GB of HBM: activations of a 1.1 B decoder at the reference batch sizes fit).  Rows at padded positions
hold zeros-in / garbage-free values that nothing reads (see csrc/softmax_attn.hip).
"""
import json
import math
import os
from types import SimpleNamespace

import torch
import torch.nn as nn


class LlamaConfig:
    """The fields of the reference's LlamaConfig (modeling_llama.py:59-187) that the decoder maths depends on."""

    def __init__(self, hidden_size=2048, intermediate_size=5632, num_hidden_layers=22, num_attention_heads=32,
                 num_key_value_heads=None, vocab_size=32000, max_position_embeddings=2048, rms_norm_eps=1e-5,
                 rope_theta=10000.0, hidden_act="silu", pad_token_id=None, attention_bias=None, **unused):
        self.hidden_size, self.intermediate_size = hidden_size, intermediate_size
        self.num_hidden_layers, self.num_attention_heads = num_hidden_layers, num_attention_heads
        self.num_key_value_heads = num_key_value_heads or num_attention_heads
        self.vocab_size, self.max_position_embeddings = vocab_size, max_position_embeddings
        self.rms_norm_eps, self.rope_theta, self.hidden_act = rms_norm_eps, rope_theta, hidden_act
        self.pad_token_id = pad_token_id
        self.model_type = unused.get("model_type", "llama")
        # Qwen2 (the reference's Pixel8M user tower, modeling_qwen2.py) is this decoder with biases on q / k / v
        self.attention_bias = (self.model_type == "qwen2") if attention_bias is None else bool(attention_bias)
        if hidden_act != "silu":
            raise NotImplementedError("only hidden_act='silu' (SwiGLU) is implemented in the fused kernels")
        if unused.get("rope_scaling"):
            raise NotImplementedError("rope_scaling is not implemented")
        if unused.get("pretraining_tp", 1) != 1:
            raise NotImplementedError("pretraining_tp > 1 only re-slices the same GEMMs; use 1")

    @classmethod
    def from_pretrained(cls, path):
        with open(os.path.join(path, "config.json")) as f:
            return cls(**json.load(f))


class LlamaRMSNorm(nn.Module):
    def __init__(self, hidden_size, eps=1e-6):
        super().__init__()
        self.weight = nn.Parameter(torch.ones(hidden_size))
        self.variance_epsilon = eps


class LlamaAttention(nn.Module):
    def __init__(self, config):
        super().__init__()
        D, nh, nkv = config.hidden_size, config.num_attention_heads, config.num_key_value_heads
        if D % nh:
            raise ValueError(f"hidden_size must be divisible by num_heads (got {D} and {nh})")
        hd = D // nh
        self.q_proj = nn.Linear(D, nh * hd, bias=config.attention_bias)
        self.k_proj = nn.Linear(D, nkv * hd, bias=config.attention_bias)
        self.v_proj = nn.Linear(D, nkv * hd, bias=config.attention_bias)
        self.o_proj = nn.Linear(nh * hd, D, bias=False)


class LlamaMLP(nn.Module):
    def __init__(self, config):
        super().__init__()
        self.gate_proj = nn.Linear(config.hidden_size, config.intermediate_size, bias=False)
        self.up_proj = nn.Linear(config.hidden_size, config.intermediate_size, bias=False)
        self.down_proj = nn.Linear(config.intermediate_size, config.hidden_size, bias=False)


class LlamaDecoderLayer(nn.Module):
    def __init__(self, config):
        super().__init__()
        self.self_attn = LlamaAttention(config)
        self.mlp = LlamaMLP(config)
        self.input_layernorm = LlamaRMSNorm(config.hidden_size, eps=config.rms_norm_eps)
        self.post_attention_layernorm = LlamaRMSNorm(config.hidden_size, eps=config.rms_norm_eps)


class LlamaModel(nn.Module):
    layer_class = LlamaDecoderLayer

    def __init__(self, config):
        super().__init__()
        self.config = config
        self.embed_tokens = nn.Embedding(config.vocab_size, config.hidden_size, config.pad_token_id)
        self.layers = nn.ModuleList([self.layer_class(config) for _ in range(config.num_hidden_layers)])
        self.norm = LlamaRMSNorm(config.hidden_size, eps=config.rms_norm_eps)
        self.gradient_checkpointing = False
        self._rope = None
        self._w_cache = {}
        for m in self.modules():                       # reference `_init_weights`: N(0, initializer_range = 0.02)
            if isinstance(m, (nn.Linear, nn.Embedding)):
                nn.init.normal_(m.weight, mean=0.0, std=0.02)
                if getattr(m, "bias", None) is not None:
                    nn.init.zeros_(m.bias)

    def get_input_embeddings(self):
        return self.embed_tokens

    def train(self, mode=True):
        self._w_cache = {}
        return super().train(mode)

    def _rope_tables(self, n_pos, device):
        """cos / sin [n_pos, head_dim/2] fp32 (modeling_llama.py:332-344), grown on demand."""
        if self._rope is None or self._rope[0].shape[0] < n_pos or self._rope[0].device != device:
            hd = self.config.hidden_size // self.config.num_attention_heads
            n_pos = max(n_pos, self.config.max_position_embeddings)
            inv_freq = 1.0 / (self.config.rope_theta ** (torch.arange(0, hd, 2, dtype=torch.float32, device=device) / hd))
            f = torch.outer(torch.arange(n_pos, dtype=torch.float32, device=device), inv_freq)
            self._rope = (f.cos().contiguous(), f.sin().contiguous())
        return self._rope

    @staticmethod
    def _qkv_params(attn):
        """([weights whose rows stack to the [q; k; v] projection], [their biases] or None).  Llama / Mistral / Qwen2 keep
        three Linears; Baichuan keeps them packed in one (`W_pack`, baichuan/modeling_baichuan.py:224)."""
        if hasattr(attn, "W_pack"):
            return [attn.W_pack.weight], None
        ws = [attn.q_proj.weight, attn.k_proj.weight, attn.v_proj.weight]
        bs = [attn.q_proj.bias, attn.k_proj.bias, attn.v_proj.bias] if attn.q_proj.bias is not None else None
        return ws, bs

    def _fused_weights(self, i, layer):
        """[q;k;v] and [gate;up] as single GEMM operands.  In training they are re-concatenated from the fp32 masters every
        step (autograd splits the gradient back); under no_grad in eval mode the bf16 copies are built once."""
        a, m = layer.self_attn, layer.mlp
        ws, bs = self._qkv_params(a)
        b_qkv = torch.cat(bs) if bs is not None else None
        w_qkv = ws[0] if len(ws) == 1 else torch.cat(ws, 0)
        if self.training or torch.is_grad_enabled():
            return (w_qkv, b_qkv, torch.cat([m.gate_proj.weight, m.up_proj.weight], 0), None)
        hit = self._w_cache.get(i)
        if hit is None:
            bf = torch.bfloat16
            hit = self._w_cache[i] = (w_qkv.to(bf), b_qkv,
                                      torch.cat([m.gate_proj.weight, m.up_proj.weight], 0).to(bf),
                                      (a.o_proj.weight.to(bf), m.down_proj.weight.to(bf)))
        return hit

    def forward(self, input_ids=None, attention_mask=None, cu_input_lens=None, position_ids=None, inputs_embeds=None,
                max_input_len=None, **unused):
        from mhr_amd.optim import fused_views
        from REC.model.hstu_functional import FusedParamLinearFn, SplitKLinearFn
        from REC.model.llm_functional import AddRMSNormFn, RMSNormFn, RopeAttentionFn, SwigluFn
        if (input_ids is None) == (inputs_embeds is None):
            raise ValueError("You have to specify exactly one of input_ids / inputs_embeds")
        if inputs_embeds is None:
            inputs_embeds = self.embed_tokens(input_ids)
        if not inputs_embeds.is_cuda:
            raise RuntimeError("the decoder runs on the MI355X only (no CPU path); move the inputs to the GPU")
        cfg = self.config
        B, L, D = inputs_embeds.shape
        nh, nkv = cfg.num_attention_heads, cfg.num_key_value_heads
        hd = D // nh
        dev = inputs_embeds.device
        x = inputs_embeds.reshape(B * L, D).float()
        key_valid = cu = positions = None
        if cu_input_lens is not None:                  # one packed row of concatenated sequences (flash_self_attn.py:61-130)
            if B != 1:
                raise ValueError("cu_input_lens describes ONE packed row of sequences (batch size 1)")
            lens = cu_input_lens.to(torch.int32)
            cu = torch.cat([lens.new_zeros(1), torch.cumsum(lens, 0, dtype=torch.int32)]).contiguous()
            n_seqs = lens.numel()
            max_len = int(max_input_len) if max_input_len is not None else int(lens.max().item())
        else:
            n_seqs, max_len = B, L
            if attention_mask is not None:
                key_valid = attention_mask.reshape(-1).to(torch.uint8).contiguous()
        if position_ids is not None:
            positions = position_ids.reshape(-1).to(torch.int32).contiguous()
            n_pos = cfg.max_position_embeddings
        else:
            n_pos = max(L, 1)
            if cu is not None:
                raise ValueError("packed sequences need explicit position_ids")
        cos, sin = self._rope_tables(n_pos, dev)
        scale = 1.0 / math.sqrt(hd)

        branch = None
        for i, layer in enumerate(self.layers):
            fv_qkv = fv_gu = None
            qkv_ws, qkv_bs = self._qkv_params(layer.self_attn)
            if torch.is_grad_enabled() and qkv_bs is None:
                # parameters laid out back to back by the fused optimizer: bf16 shadow view in, flat gradient view out
                m_ = layer.mlp
                fv_qkv = fused_views(qkv_ws)
                fv_gu = fused_views([m_.gate_proj.weight, m_.up_proj.weight])
            if fv_qkv is not None and fv_gu is not None:
                w_qkv = b_qkv = w_gu = cached = None
            else:
                fv_qkv = fv_gu = None
                w_qkv, b_qkv, w_gu, cached = self._fused_weights(i, layer)
            pre = cached is not None
            if branch is None:
                h = RMSNormFn.apply(x, layer.input_layernorm.weight, layer.input_layernorm.variance_epsilon)
            else:
                x, h = AddRMSNormFn.apply(x, branch, layer.input_layernorm.weight, layer.input_layernorm.variance_epsilon)
            if fv_qkv is not None:
                qkv = FusedParamLinearFn.apply(h, fv_qkv[0], fv_qkv[1], qkv_ws[0])
            else:
                qkv = SplitKLinearFn.apply(h, w_qkv, b_qkv, True, w_qkv if pre else None)
            a = RopeAttentionFn.apply(qkv, cos, sin, positions, key_valid, cu, n_seqs, max_len, nh, nkv, hd, scale)
            o = SplitKLinearFn.apply(a, layer.self_attn.o_proj.weight, None, True, cached[0] if pre else None)
            x, h = AddRMSNormFn.apply(x, o, layer.post_attention_layernorm.weight, layer.post_attention_layernorm.variance_epsilon)
            if fv_gu is not None:
                gu = FusedParamLinearFn.apply(h, fv_gu[0], fv_gu[1], layer.mlp.gate_proj.weight)
            else:
                gu = SplitKLinearFn.apply(h, w_gu, None, True, w_gu if pre else None)
            act = SwigluFn.apply(gu)
            branch = SplitKLinearFn.apply(act, layer.mlp.down_proj.weight, None, True, cached[1] if pre else None)
        if branch is None:
            y = RMSNormFn.apply(x, self.norm.weight, self.norm.variance_epsilon)
        else:
            _, y = AddRMSNormFn.apply(x, branch, self.norm.weight, self.norm.variance_epsilon)
        hidden = y.view(B, L, D)
        return SimpleNamespace(last_hidden_state=hidden, hidden_states=(hidden,), past_key_values=None, attentions=None)


class LlamaForCausalLM(nn.Module):
    model_class = LlamaModel
    config_class = LlamaConfig

    def __init__(self, config):
        super().__init__()
        self.config = config
        self.model = self.model_class(config)
        self.vocab_size = config.vocab_size
        self.lm_head = nn.Linear(config.hidden_size, config.vocab_size, bias=False)
        nn.init.normal_(self.lm_head.weight, mean=0.0, std=0.02)

    @classmethod
    def from_pretrained(cls, path, config=None):
        """Weights from `<path>/model.safetensors` (or `pytorch_model.bin`, loaded with weights_only=True)."""
        config = config or cls.config_class.from_pretrained(path)
        model = cls(config)
        st = os.path.join(path, "model.safetensors")
        if os.path.exists(st):
            from safetensors.torch import load_file
            state = load_file(st)
        else:
            state = torch.load(os.path.join(path, "pytorch_model.bin"), map_location="cpu", weights_only=True)
        model.load_state_dict(state, strict=False)
        return model

    def get_input_embeddings(self):
        return self.model.embed_tokens

    def forward(self, input_ids=None, attention_mask=None, position_ids=None, inputs_embeds=None, cu_input_lens=None,
                emb_tokens=None, emb_pos=None, emb_token_n=0, max_input_len=None, **unused):
        if inputs_embeds is None:
            inputs_embeds = self.model.embed_tokens(input_ids)                       # [T, D] for the packed item tower
            if emb_token_n > 0:                      # the learnable [ITEM] token replaces the last token of every item
                idx = (emb_pos - 1).long().unsqueeze(1).expand(-1, emb_tokens.size(-1))
                inputs_embeds = inputs_embeds.scatter(0, idx, emb_tokens.view(1, -1).expand(emb_pos.numel(), -1).to(inputs_embeds.dtype))
            if inputs_embeds.dim() == 2:
                inputs_embeds = inputs_embeds.unsqueeze(0)
        out = self.model(attention_mask=attention_mask, position_ids=position_ids, inputs_embeds=inputs_embeds,
                         cu_input_lens=cu_input_lens, max_input_len=max_input_len)
        out.logits = None                            # the reference computes lm_head logits here and never reads them
        return out
