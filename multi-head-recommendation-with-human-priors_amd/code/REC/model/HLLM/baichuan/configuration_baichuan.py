"""Baichuan2 decoder configuration: the fields of the reference's `BaichuanConfig`
(`code/REC/model/HLLM/baichuan/configuration_baichuan.py:29-69`) that the decoder maths reads."""
from REC.model.HLLM.modeling_llama import LlamaConfig


class BaichuanConfig(LlamaConfig):
    model_type = "baichuan"

    def __init__(self, vocab_size=125696, hidden_size=4096, intermediate_size=11008, num_hidden_layers=32,
                 num_attention_heads=32, hidden_act="silu", max_position_embeddings=4096, rms_norm_eps=1e-6, pad_token_id=0,
                 **unused):
        for k in ("num_key_value_heads", "rope_theta", "attention_bias", "model_type"):
            unused.pop(k, None)       # full multi-head attention, rotary base 10000 (modeling_baichuan.py:136-139), no biases
        super().__init__(hidden_size=hidden_size, intermediate_size=intermediate_size, num_hidden_layers=num_hidden_layers,
                         num_attention_heads=num_attention_heads, num_key_value_heads=num_attention_heads,
                         vocab_size=vocab_size, max_position_embeddings=max_position_embeddings, rms_norm_eps=rms_norm_eps,
                         rope_theta=10000.0, hidden_act=hidden_act, pad_token_id=pad_token_id, attention_bias=False,
                         model_type="baichuan", **unused)
