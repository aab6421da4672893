"""Baichuan2 decoder for the HLLM towers (BASELINE.json configs[4]: Baichuan2-7B), MI355X-native.

Module tree and `state_dict` names of the reference `code/REC/model/HLLM/baichuan/modeling_baichuan.py`:
`model.embed_tokens`, `model.layers.{i}.self_attn.W_pack` (ONE Linear for q|k|v, line 224), `.self_attn.o_proj`,
`.mlp.{gate,down,up}_proj`, `.input_layernorm`, `.post_attention_layernorm`, `model.norm`, `lm_head.weight` (NormHead,
600-625: only its weight exists; the path never reads the logits).

The maths is the Llama block (RMSNorm 110-133, rotary base 10000 with rotate_half 136-180, causal softmax attention
301-334, SwiGLU 183-209) with as many KV heads as query heads, so the forward is `LlamaModel.forward`: the fused
RMSNorm / RoPE / attention / SwiGLU gfx950 kernels, with `W_pack` already being the single [q; k; v] GEMM operand the
Llama path has to concatenate (bf16 shadow view in, flat gradient view out under the fused optimizer).  Beyond the
reference's signature the causal-LM wrapper also accepts the packed item-tower arguments (`cu_input_lens`,
`emb_tokens`, `emb_pos`, `emb_token_n`) of the Llama wrapper.
"""
import torch
import torch.nn as nn

from REC.model.HLLM.baichuan.configuration_baichuan import BaichuanConfig
from REC.model.HLLM.modeling_llama import LlamaForCausalLM, LlamaMLP, LlamaModel, LlamaRMSNorm


class RMSNorm(LlamaRMSNorm):
    pass


class MLP(LlamaMLP):
    pass


class Attention(nn.Module):
    def __init__(self, config):
        super().__init__()
        D, nh = config.hidden_size, config.num_attention_heads
        if D % nh:
            raise ValueError(f"hidden_size must be divisible by num_heads (got `hidden_size`: {D} and `num_heads`: {nh}).")
        self.W_pack = nn.Linear(D, 3 * D, bias=False)
        self.o_proj = nn.Linear(D, D, bias=False)


class DecoderLayer(nn.Module):
    def __init__(self, config):
        super().__init__()
        self.self_attn = Attention(config)
        self.mlp = MLP(config)
        self.input_layernorm = RMSNorm(config.hidden_size, eps=config.rms_norm_eps)
        self.post_attention_layernorm = RMSNorm(config.hidden_size, eps=config.rms_norm_eps)


class BaichuanModel(LlamaModel):
    layer_class = DecoderLayer


class NormHead(nn.Module):
    """Holder of `lm_head.weight` [vocab, hidden] (modeling_baichuan.py:600-606); its normalised logits are not part of
    the recommendation path."""

    def __init__(self, hidden_size, vocab_size, bias=False):
        super().__init__()
        self.weight = nn.Parameter(torch.empty(vocab_size, hidden_size))
        nn.init.kaiming_uniform_(self.weight, a=5 ** 0.5)


class BaichuanForCausalLM(LlamaForCausalLM):
    model_class = BaichuanModel
    config_class = BaichuanConfig

    def __init__(self, config):
        nn.Module.__init__(self)
        self.config = config
        self.model = BaichuanModel(config)
        self.vocab_size = config.vocab_size
        self.lm_head = NormHead(config.hidden_size, config.vocab_size, bias=False)
