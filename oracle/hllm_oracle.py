"""CPU restatement (torch, fp32) of the HLLM twin: the Llama-style decoder blocks and the HLLM glue around the
shared multi-head loss / predict code.  TEST INFRASTRUCTURE ONLY - imported by tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg, never by the product path.

What pins it:
  * decoder blocks: tests/golden/llama_decoder_*.npz, produced by tests/gen_golden.py from the reference's own
    `REC/model/HLLM/modeling_llama.py` (LlamaForCausalLM, eager attention path, random-init tiny config);
  * heads / loss / predict: the HSTU fixtures - hllm.py:506-763, 785-883 is the same code as hstu.py:648-872, 915-1016
     and both go through hstu_oracle.multihead_loss / predict_scores here;
  * HLLM-specific glue (hllm.py:482-502, 527-547, 626-646, 771-783): restated from the source text, PARITY UNPINNED -
    `REC.model.HLLM.hllm` cannot be imported in the build container (unconditional flash_attn imports);
  * packed `cu_input_lens` attention (flash_self_attn.py:61-130): restated, PARITY UNPINNED (needs flash-attn).

All file:line references are under /root/reference/code/REC/model/HLLM/.
"""
import math

import torch

from . import hstu_oracle as HO


# --------------------------------------------------------------------------- #
# Llama decoder blocks
# --------------------------------------------------------------------------- #
def rms_norm(x, w, eps):
    """modeling_llama.py:266-280"""
    var = x.float().pow(2).mean(-1, keepdim=True)
    return w * (x.float() * torch.rsqrt(var + eps))


def rope_tables(head_dim, max_pos, theta=10000.0):
    """cos / sin [max_pos, head_dim/2] of position * theta^(-2i/head_dim).  modeling_llama.py:332-344 with the default
    rope init (modeling_rope_utils.py `_compute_default_rope_parameters`)."""
    inv_freq = 1.0 / (theta ** (torch.arange(0, head_dim, 2, dtype=torch.float32) / head_dim))
    f = torch.outer(torch.arange(max_pos, dtype=torch.float32), inv_freq)
    return f.cos(), f.sin()


def apply_rope(x, cos, sin, position_ids):
    """x [B, heads, L, hd]; rotate_half convention.  modeling_llama.py:426-441"""
    c = torch.cat([cos, cos], -1)[position_ids][:, None]        # [B,1,L,hd]
    s = torch.cat([sin, sin], -1)[position_ids][:, None]
    h = x.shape[-1] // 2
    rot = torch.cat([-x[..., h:], x[..., :h]], dim=-1)
    return x * c + rot * s


def attention(q, k, v, allowed):
    """softmax(q k^T / sqrt(hd) restricted to `allowed` [B,1|H,L,L] bool) v.  modeling_llama.py:648-682.  Rows with no
    admissible key return zeros (the reference's eager path averages over an arbitrary finite set there - its additive
    finfo.min masks - and nothing reads those rows; see softmax_attn.hip)."""
    hd = q.shape[-1]
    s = (q @ k.transpose(-1, -2)) / math.sqrt(hd)
    s = s.masked_fill(~allowed, float("-inf"))
    dead = ~allowed.any(-1, keepdim=True)
    p = torch.softmax(s.masked_fill(dead, 0.0), dim=-1).masked_fill(dead, 0.0)
    return HO._r(p) @ v                         # (under HO.MIXED the probability tile is a bf16 matrix operand)


def decoder_layer(x, w, pfx, lcfg, allowed, cos, sin, position_ids):
    """modeling_llama.py:729-795 (pre-norm attention + SwiGLU MLP, residual adds)."""
    B, L, D = x.shape
    nh, nkv = lcfg["num_attention_heads"], lcfg["num_key_value_heads"]
    hd = D // nh
    h = rms_norm(x, w[pfx + "input_layernorm.weight"], lcfg["rms_norm_eps"])
    def proj(name):                # Qwen2 towers carry q/k/v biases (modeling_qwen2.py), Llama / Mistral do not
        y = HO._mm(h, w[pfx + f"self_attn.{name}_proj.weight"].T)
        return HO._r(y + HO._r(w[pfx + f"self_attn.{name}_proj.bias"])) if pfx + f"self_attn.{name}_proj.bias" in w else y

    q = proj("q").view(B, L, nh, hd).transpose(1, 2)
    k = proj("k").view(B, L, nkv, hd).transpose(1, 2)
    v = proj("v").view(B, L, nkv, hd).transpose(1, 2)
    q, k = HO._r(apply_rope(q, cos, sin, position_ids)), HO._r(apply_rope(k, cos, sin, position_ids))
    k = k.repeat_interleave(nh // nkv, dim=1)                   # repeat_kv, modeling_llama.py:489-500
    v = v.repeat_interleave(nh // nkv, dim=1)
    a = HO._r(attention(q, k, v, allowed)).transpose(1, 2).reshape(B, L, D)
    x = x + HO._mm(a, w[pfx + "self_attn.o_proj.weight"].T)
    h = rms_norm(x, w[pfx + "post_attention_layernorm.weight"], lcfg["rms_norm_eps"])
    g = HO._mm(h, w[pfx + "mlp.gate_proj.weight"].T)
    u = HO._mm(h, w[pfx + "mlp.up_proj.weight"].T)
    return x + HO._mm(HO.silu(g) * u, w[pfx + "mlp.down_proj.weight"].T)


def llama_decoder(w, lcfg, inputs_embeds, attention_mask=None, position_ids=None, seg_ids=None, prefix="model."):
    """LlamaModel.forward on `inputs_embeds` [B,L,D] -> last hidden state after the final norm (what the callers take
    as `hidden_states[-1]`, modeling_llama.py:1108-1112).  attention_mask [B,L] marks valid keys (970-1048: causal AND
    key-valid); seg_ids [B,L] (optional) restricts attention to tokens of the same segment - the packed `cu_input_lens`
    batches of the item tower (flash_self_attn.py:61-130).  position_ids default to 0..L-1 (1020-1028)."""
    B, L, D = inputs_embeds.shape
    hd = D // lcfg["num_attention_heads"]
    if position_ids is None:
        position_ids = torch.arange(L)[None].expand(B, L)
    cos, sin = rope_tables(hd, int(position_ids.max()) + 1, lcfg.get("rope_theta", 10000.0))
    allowed = torch.tril(torch.ones(L, L, dtype=torch.bool))[None, None].expand(B, 1, L, L)
    if attention_mask is not None:
        allowed = allowed & attention_mask.bool()[:, None, None, :]
    if seg_ids is not None:
        allowed = allowed & (seg_ids[:, None, :, None] == seg_ids[:, None, None, :])
    x = inputs_embeds.float()
    for i in range(lcfg["num_hidden_layers"]):
        x = decoder_layer(x, w, f"{prefix}layers.{i}.", lcfg, allowed, cos, sin, position_ids)
    return rms_norm(x, w[prefix + "norm.weight"], lcfg["rms_norm_eps"])


def baichuan_decoder(w, lcfg, inputs_embeds, attention_mask=None, position_ids=None, seg_ids=None, prefix="model."):
    """BaichuanModel.forward (REC/model/HLLM/baichuan/modeling_baichuan.py:465-597) on `inputs_embeds`: the Llama block
    structure (DecoderLayer 341-395 = RMSNorm -> attention -> residual -> RMSNorm -> SwiGLU MLP -> residual; RMSNorm
    110-133; rotary embedding with base 10000 and rotate_half 136-180; softmax attention 301-334) with the q, k, v
    projections stored as ONE `W_pack` Linear whose output is cut into thirds (224, 258-262) and as many KV heads as
    query heads.  Restated by viewing W_pack's rows [0,D) / [D,2D) / [2D,3D) as q_proj / k_proj / v_proj."""
    D = lcfg["hidden_size"]
    view = dict(w)
    for i in range(lcfg["num_hidden_layers"]):
        pack = w[f"{prefix}layers.{i}.self_attn.W_pack.weight"]
        for j, name in enumerate(("q_proj", "k_proj", "v_proj")):
            view[f"{prefix}layers.{i}.self_attn.{name}.weight"] = pack[j * D:(j + 1) * D]
    cfg = dict(lcfg, num_key_value_heads=lcfg["num_attention_heads"], rope_theta=10000.0)
    return llama_decoder(view, cfg, inputs_embeds, attention_mask, position_ids, seg_ids, prefix)


def dummy_llm(w, prefix, inputs_embeds):
    """REC/model/layers.py:94-111 DummyLLM: one Linear on the input embeddings."""
    return inputs_embeds @ w[prefix + "embed_layer.weight"].T + w[prefix + "embed_layer.bias"]


# --------------------------------------------------------------------------- #
# HLLM glue (frozen item tower: embeddings come from `all_item_embeds`)
# --------------------------------------------------------------------------- #
def user_encoder(w, cfg, embs, key_mask):
    if cfg.get("dummy_llm"):
        return dummy_llm(w, "user_llm.", embs)
    if cfg["user_llm_config"].get("model_type") == "baichuan":
        return baichuan_decoder(w, cfg["user_llm_config"], embs, key_mask, prefix="user_llm.model.")
    return llama_decoder(w, cfg["user_llm_config"], embs, key_mask, prefix="user_llm.model.")


def train_forward(w, cfg, interaction, all_item_embeds):
    """HLLM.forward(mode='train') with `freeze_item_llm`.  hllm.py:476-763.

    interaction: dict with attention_mask [B,L+P], pos_item_ids [B,L+P], pos_tag_categories [B,L+P,C],
    neg_item_ids [B,n_neg] and/or neg_item_ids_cat{c} [B,n_neg].  `w` holds the user decoder under `user_llm.` and the
    heads / logit_scale under the HSTU names (same module names in hllm.py:137-243)."""
    mask = interaction["attention_mask"].bool()
    P = cfg["pred_len"]
    e = all_item_embeds[interaction["pos_item_ids"]].float()                # hllm.py:486-488
    out = user_encoder(w, cfg, e[:, :-P], mask[:, :-P])                    # hllm.py:501-502

    def negs_for(pool):            # shared pool (hllm.py:528-531, 545) is asked for as -1, category pools (627-630, 645) as c
        ids = interaction["neg_item_ids"] if pool == -1 else interaction[f"neg_item_ids_cat{pool}"]
        n = all_item_embeds[ids].float()
        return HO.l2n(n).reshape(-1, n.shape[-1])

    return HO.multihead_loss(w, cfg, e, out, mask, interaction.get("pos_tag_categories"), negs_for, 2)


def predict_scores(w, cfg, item_seq, all_item_feature, all_item_tags, target_tags, logs=None):
    """HLLM.predict.  hllm.py:765-883: embeddings of the history looked up in `all_item_feature`, user decoder, last
    position, then the same heads / masks as HSTU.predict."""
    emb = all_item_feature[item_seq].float()                                # hllm.py:773
    last = user_encoder(w, cfg, emb, item_seq > 0)[:, -1]                   # hllm.py:781-783
    return HO.predict_scores(w, cfg, item_seq, all_item_feature, all_item_tags, target_tags, logs=logs, last=last)
