"""fp32 CPU oracle of the HSTU model maths (test infrastructure only).

Functional restatement: every function takes plain tensors / a `weights` dict
keyed by the reference's `state_dict` names and a `cfg` dict keyed by the
reference's config names.  Citations are `file:line` under
/root/reference/code/REC/.

Pinned by tests/golden/hstu_*.npz (generated from the reference, see
tests/gen_golden.py).
"""
import math
from collections import OrderedDict

import torch

F32_MIN = torch.finfo(torch.float32).min


# --------------------------------------------------------------------------- #
# config helpers
# --------------------------------------------------------------------------- #
def num_heads_total(cfg):
    """H = S*C (multiplicative / hierarchical) or S+C (additive). model/IDNet/hstu.py:361-366"""
    S, C = cfg["num_segment_head"], cfg["num_prior_head"]
    hi = cfg["head_interaction"]
    if hi in ("multiplicative", "hierarchical"):
        return S * C
    if hi == "additive":
        return S + C
    raise ValueError(f"Unknown head_interaction: {hi}")


def horizon_discount(cfg):
    """lambda^p / sum_p lambda^p.  model/IDNet/hstu.py:436-438"""
    lam = torch.tensor([cfg["medusa_lambda"] ** p for p in range(cfg["pred_len"])])
    return (lam / lam.sum()).float()


def seg_len_of(cfg):
    """model/IDNet/hstu.py:428-432"""
    if cfg["medusa_num_layers"] > 0:
        return cfg["pred_len"] // cfg["num_segment_head"]
    return cfg["pred_len"]


def prior_loss_weights(cfg):
    """model/IDNet/hstu.py:503-510: category frequency or uniform 1/C."""
    C = cfg["num_prior_head"]
    if cfg["loss"] == "prior" and cfg.get("weighted_prior_loss"):
        counts = cfg["category_counts"]          # {name: count}
        to_int = cfg["category_to_int"]          # {name: int}
        tot = sum(counts.values())
        w = [0.0] * C
        for name, c in counts.items():
            w[to_int[name]] = c / tot
        return w
    return [1.0 / C] * C


# --------------------------------------------------------------------------- #
# bf16-mixed emulation (tests only)
# --------------------------------------------------------------------------- #
# The reference's GPU recipe is Fabric bf16-mixed autocast (run.py:3): matrix products take bf16 operands and emit bf16.
# With MIXED = a rounding function (e.g. tests/kernel_oracles.bf16_round, straight-through for autograd) every dense product
# of this oracle rounds its operands and its result the way the MI355X path stores them; None (default) is the fp32
# arithmetic the golden fixtures were generated with.  Set only by tests that want a tolerance tighter than "bf16-sized".
MIXED = None


def _r(t):
    return t if MIXED is None else MIXED(t)


def _mm(a, b):
    """a @ b with bf16-rounded operands and result under MIXED."""
    return a @ b if MIXED is None else MIXED(MIXED(a) @ MIXED(b))


# --------------------------------------------------------------------------- #
# encoder
# --------------------------------------------------------------------------- #
def layer_norm(x, eps=1e-6):
    """Affine-free LayerNorm over the last dim. model/IDNet/hstu.py:213-219"""
    mu = x.mean(-1, keepdim=True)
    var = ((x - mu) ** 2).mean(-1, keepdim=True)
    return (x - mu) / torch.sqrt(var + eps)


def silu(x):
    return x * torch.sigmoid(x)


def hstu_attention(q, k, v, key_valid, n_heads, operand_round=None):
    """Pointwise-gated causal attention. model/IDNet/hstu.py:137-160.

    q,k: [B,L,Hh*dqk]  v: [B,L,Hh*dv]  key_valid: [B,L] bool.
    A[b,n,h,:] = sum_{m<=n, key_valid[b,m]} silu(q[b,n,h].k[b,m,h]) / L * v[b,m,h,:]
    The normaliser is the padded length L (hstu.py:146,153), and padded *query*
    rows still produce an output.
    operand_round (tests only; None = the reference's fp32): rounding applied to the gate tile where the GPU kernel
    feeds it to its second matrix product as a bf16 operand (the caller rounds q, k, v the same way).
    """
    B, L, _ = q.shape
    dqk = q.shape[-1] // n_heads
    dv = v.shape[-1] // n_heads
    qh = q.reshape(B, L, n_heads, dqk).permute(0, 2, 1, 3)
    kh = k.reshape(B, L, n_heads, dqk).permute(0, 2, 1, 3)
    vh = v.reshape(B, L, n_heads, dv).permute(0, 2, 1, 3)
    s = silu(qh @ kh.transpose(-1, -2)) / L                      # [B,Hh,L,L]
    causal = torch.ones(L, L, dtype=torch.bool).tril()
    m = causal[None, None] & key_valid[:, None, None, :]
    s = s * m
    if operand_round is not None:
        s = operand_round(s)
    out = s @ vh                                                  # [B,Hh,L,dv]
    return out.permute(0, 2, 1, 3).reshape(B, L, n_heads * dv)


def hstu_layer(x, w_uvqk, w_o, b_o, key_valid, n_heads, act="silu", drop_mask=None, eps=1e-6):
    """One SequentialTransductionUnit. model/IDNet/hstu.py:221-290.

    x [B,L,D]; w_uvqk [D,4D] (u|v|q|k column blocks, each n_heads*d);
    w_o [D,D] (nn.Linear weight, y = a @ w_o.T + b_o).
    drop_mask: optional [B,L,D] keep-mask already scaled by 1/(1-p).
    """
    D = x.shape[-1]
    h = _mm(layer_norm(x, eps), w_uvqk)
    if act == "silu":
        h = silu(h)
    u, v, q, k = torch.split(h, [D, D, D, D], dim=-1)
    a = _r(hstu_attention(_r(q), _r(k), _r(v), key_valid, n_heads, operand_round=MIXED))
    o_in = u * layer_norm(a, eps)
    if drop_mask is not None:
        o_in = o_in * drop_mask
    return _r(_mm(o_in, w_o.T) + _r(b_o)) + x


def hstu_encoder(x, weights, cfg, key_valid):
    """model/IDNet/hstu.py:307-328"""
    for i in range(cfg["n_layers"]):
        p = f"_hstu._attention_layers.{i}."
        x = hstu_layer(x, weights[p + "_uvqk"], weights[p + "_o.weight"], weights[p + "_o.bias"],
                       key_valid, cfg["n_heads"], cfg.get("hidden_act") or "silu")
    return x


def item_tower(weights, e):
    """item_id_proj_tower: identity or bias-free Linear. model/IDNet/hstu.py:414"""
    w = weights.get("item_id_proj_tower.weight")
    return e if w is None else e @ w.T


def embed_inputs(weights, cfg, ids):
    """item gather + position add for the first L columns. model/IDNet/hstu.py:637-643, 879-884"""
    L = cfg["MAX_ITEM_LIST_LENGTH"]
    e = item_tower(weights, weights["item_embedding.weight"][ids])       # [B, L(+P), D]
    x = e[:, :L] + weights["position_embedding.weight"][:L][None]
    return e, x


# --------------------------------------------------------------------------- #
# decoding heads
# --------------------------------------------------------------------------- #
def _affine_ln(x, w, b, eps=1e-5):
    mu = x.mean(-1, keepdim=True)
    var = ((x - mu) ** 2).mean(-1, keepdim=True)
    return (x - mu) / torch.sqrt(var + eps) * w + b


def res_block(x, weights, prefix):
    """x + silu(Linear(x)), optional affine LayerNorm first. model/llm_heads.py:5-40"""
    if prefix + "norm.weight" in weights:
        x = _affine_ln(x, weights[prefix + "norm.weight"], weights[prefix + "norm.bias"])
    return x + silu(_r(_mm(x, weights[prefix + "linear.weight"].T) + _r(weights[prefix + "linear.bias"])))


def _sequential(x, weights, prefix):
    """Apply an nn.Sequential whose children are LayerNorm/Linear/SiLU/ResBlock,
    recognised by the parameter names present under `prefix`."""
    idxs = sorted({int(k[len(prefix):].split(".")[0]) for k in weights if k.startswith(prefix)})
    if not idxs:
        return x
    last = max(idxs)
    for i in range(last + 1):
        p = f"{prefix}{i}."
        if p + "linear.weight" in weights:
            x = res_block(x, weights, p)
        elif p + "weight" in weights and weights[p + "weight"].dim() == 2:
            x = x @ weights[p + "weight"].T + weights[p + "bias"]
        elif p + "weight" in weights:
            x = _affine_ln(x, weights[p + "weight"], weights[p + "bias"])
        else:                                   # parameter-free child: the SiLU of the bottleneck
            x = silu(x)
    return x


def tie_repeated_resblocks(weights):
    """hstu.py:486-493 builds `nn.Sequential(*([ResBlock(D)] * n))`: ONE ResBlock applied n times,
    exported under n aliasing state_dict keys.  Re-alias entries `medusa_head.{h}.{i}.*` (i>0) to
    `medusa_head.{h}.0.*` so autograd on the oracle accumulates into a single tensor as the reference does."""
    for k in list(weights):
        parts = k.split(".")
        if parts[0] == "medusa_head" and len(parts) >= 4 and parts[2].isdigit() and parts[2] != "0":
            base = ".".join(parts[:2] + ["0"] + parts[3:])
            if base in weights:
                weights[k] = weights[base]
    return weights


def decode_heads(weights, cfg, x):
    """All decoding heads applied to x [..., D] -> [..., H, D] (H axis inserted before D).

    multiplicative / additive: H independent stacks, the *same* ResBlock applied
    medusa_num_layers times (hstu.py:486-493: the list `[ResBlock]*n` repeats one
    instance, so state_dict keys `.0.` ... `.{n-1}.` alias one tensor).
    hierarchical (hstu.py:443-484, 652-663): cat head c, then + segment_emb[s],
    then seg head [c][s]; head index h = s*C + c.
    """
    S, C = cfg["num_segment_head"], cfg["num_prior_head"]
    H = num_heads_total(cfg)
    nl = cfg["medusa_num_layers"]
    outs = []
    if nl == 0:
        outs = [x for _ in range(H)]
    elif cfg["head_interaction"] == "hierarchical":
        cat = [_sequential(x, weights, f"medusa_cat_head.{c}.") for c in range(C)]
        for s in range(S):
            for c in range(C):
                h_in = cat[c]
                if "segment_emb.weight" in weights:
                    h_in = h_in + weights["segment_emb.weight"][s]
                outs.append(_sequential(h_in, weights, f"medusa_seg_head.{c}.{s}."))
    else:
        for h in range(H):
            outs.append(_sequential(x, weights, f"medusa_head.{h}."))
    return torch.stack(outs, dim=-2)


# --------------------------------------------------------------------------- #
# sampled softmax
# --------------------------------------------------------------------------- #
def l2n(x):
    return x / x.norm(dim=-1, keepdim=True)


def nce_logits(q, p, negs_n, logit_scale, thres, operand_round=None):
    """model/IDNet/hstu.py:600-619.  q,p: [T,D] raw; negs_n: [Nn,D] already L2-normalised.
    operand_round (tests only; None = the reference's fp32): rounding applied to the normalised rows, which the GPU kernels
    feed to the matrix cores as bf16 operands.

    Returns logits [T, 1+Nn] (label 0).  False negatives (cos(p, neg) > thres) are
    suppressed to -inf (finfo.min * exp(scale) overflows to -inf, as in the reference).
    logit_scale is clamped to [0, ln 100] (in the reference the clamp is in-place on the parameter).
    """
    scale = torch.clamp(logit_scale, 0.0, math.log(100.0)).exp()
    qn, pn = l2n(q), l2n(p)
    if operand_round is not None:
        qn, pn = operand_round(qn), operand_round(pn)
    pos = (qn * pn).sum(-1, keepdim=True)
    neg = qn @ negs_n.T
    fix = pn @ negs_n.T
    neg = neg.masked_fill(fix > thres, F32_MIN)
    return torch.cat([pos, neg], dim=-1) * scale


def token_ce(logits):
    """F.cross_entropy(logits, 0, reduction='none'). hstu.py:697"""
    return torch.logsumexp(logits, dim=-1) - logits[:, 0]


def train_topk_logs(logits):
    """hstu.py:621-629 (labels are all 0)."""
    out = OrderedDict()
    out["nce_samples"] = (logits > F32_MIN / 100).sum(dim=1).float().mean()
    for k in (1, 5, 10, 50, 100):
        if k > logits.shape[-1]:
            break
        idx = logits.topk(k, dim=-1).indices
        out[f"nce_top{k}_acc"] = (idx == 0).any(dim=-1).float().mean()
    return out


def future_targets(e, mask, tags, L, P):
    """Windows of the next P items for every position. hstu.py:682-685, 743-746, 808-809.

    tgt[b,p,l] = e[b, l+1+p];  valid[b,p,l] = mask[b,l] & mask[b,l+1+p];
    tag[b,p,l,c] = tags[b,l+1+p,c]
    """
    B = e.shape[0]
    idx = torch.arange(L)[None, :] + 1 + torch.arange(P)[:, None]       # [P,L]
    tgt = e[:, idx]                                                        # [B,P,L,D]
    valid = mask[:, None, :L] & mask[:, idx]                               # [B,P,L]
    tg = tags[:, idx] if tags is not None and tags.numel() > 0 else None  # [B,P,L,C]
    return tgt, valid, tg


def _per_offset_mean(tok_loss, p_of_tok, P):
    s = torch.zeros(P).index_add_(0, p_of_tok, tok_loss)
    n = torch.zeros(P).index_add_(0, p_of_tok, torch.ones_like(tok_loss))
    return s / n.clamp_min(1.0)


def asymmetric_loss(x, y, gamma_pos, gamma_neg, clip=0.05, eps=1e-8):
    """REC/model/layers.py:16-83 (ASL): summed over the last dim, mean over the rest."""
    xs_pos = torch.sigmoid(x)
    xs_neg = 1 - xs_pos
    if clip is not None and clip > 0:
        xs_neg = (xs_neg + clip).clamp(max=1)
    loss = y * torch.log(xs_pos.clamp(min=eps)) + (1 - y) * torch.log(xs_neg.clamp(min=eps))
    if gamma_neg > 0 or gamma_pos > 0:
        pt = xs_pos * y + xs_neg * (1 - y)
        loss = loss * torch.pow(1 - pt, gamma_pos * y + gamma_neg * (1 - y))
    return (-loss.sum(dim=-1)).mean()


def switch_logits(weights, cfg, out, head_out, c):
    """aux_cat_head[c] on the encoder output ('in') or on [encoder output | head c output] ('in_out').
    out [..., D]; head_out [..., H, D] (head axis second to last).  model/IDNet/hstu.py:770-784, 940-947."""
    S = cfg["num_segment_head"]
    if cfg["prior_switch"] == "in":
        aux_in = out
    else:
        h = S + c if cfg["head_interaction"] == "additive" else c
        aux_in = torch.cat([out, head_out[..., h, :]], dim=-1)
    if cfg.get("detach_aux_in"):
        aux_in = aux_in.detach()
    return (aux_in @ weights[f"aux_cat_head.{c}.weight"].T + weights[f"aux_cat_head.{c}.bias"]).squeeze(-1)


def prior_switch_loss(weights, cfg, out, heads, tg, c, w_c):
    """Per-category auxiliary "will the next pred_len items contain category c" head.  model/IDNet/hstu.py:757-805.
    out [B,L,D]; heads [B,H,L,D]; tg [B,P,L,C] future tags.  Returns (loss, accuracy)."""
    target = tg[..., c].bool().any(dim=1).float()                      # [B,L]
    logits = switch_logits(weights, cfg, out, heads.permute(0, 2, 1, 3), c)   # [B,L]
    if cfg.get("switch_last_only"):
        target, logits = target[:, -1:], logits[:, -1:]
    if cfg.get("asym_switch_loss"):
        loss = asymmetric_loss(logits, target, cfg.get("gamma_pos", 4.0), cfg.get("gamma_neg", 0.0))
    else:
        p = max(min(float(w_c[c]), 1.0 - 1e-6), 1e-6)
        loss = torch.nn.functional.binary_cross_entropy_with_logits(logits, target, pos_weight=torch.tensor((1.0 - p) / p))
    acc = ((logits >= 0).int() == target.int()).float().mean()
    return loss, acc.detach()


def gather_negatives(weights, neg_ids):
    """hstu.py:669-673 / 751-755 at world size 1: gather, L2-normalise, flatten."""
    n = item_tower(weights, weights["item_embedding.weight"][neg_ids])
    return _r(l2n(n)).reshape(-1, n.shape[-1])            # (the negatives are stored normalised in bf16 under MIXED)


def train_forward(weights, cfg, batch, extra_negs=None):
    """HSTU.forward for loss in {nce, prior}.  model/IDNet/hstu.py:631-872.

    batch = (items [B,L+P] i64, neg_items [B,pools,n_neg] i64, mask [B,L+P], tags [B,L+P,C] or empty).
    extra_negs: optional dict pool_index -> [M,D] already-normalised negatives from
    other ranks (appended after the local ones, mirroring all_gather's rank order for rank 0).
    Returns an OrderedDict with 'loss' and the reference's logging scalars; dropout is off.
    """
    items, neg_items, mask, tags = batch
    mask = mask.bool()
    L, P = cfg["MAX_ITEM_LIST_LENGTH"], cfg["pred_len"]
    S, C = cfg["num_segment_head"], cfg["num_prior_head"]
    B = items.shape[0]
    hi = cfg["head_interaction"]
    loss_kind = cfg["loss"]
    by_cat = bool(cfg.get("neg_sample_by_cat")) and loss_kind == "prior"
    thres = cfg.get("nce_thres") or 0.99
    lam = horizon_discount(cfg)
    names = cfg.get("int_to_category") or {c: str(c) for c in range(C)}

    e, x = embed_inputs(weights, cfg, items)
    out = hstu_encoder(x, weights, cfg, mask[:, :L])

    def negs_for(pool):
        n = gather_negatives(weights, neg_items[:, pool])
        if extra_negs is not None and pool in extra_negs:
            n = torch.cat([n, extra_negs[pool]], 0)
        return n

    return multihead_loss(weights, cfg, e, out, mask, tags, negs_for, neg_items.shape[1])


def multihead_loss(weights, cfg, e, out, mask, tags, negs_for, n_pools):
    """Everything of the training forward after the sequence encoder: decoding heads, future windows, sampled softmax per
    head / prior category, per-offset weighting, prior switch.  model/IDNet/hstu.py:648-872 and, line for line the same
    code, model/HLLM/hllm.py:506-763.  e [B,L+P,D] item embeddings (targets), out [B,L,D] encoder output, mask [B,L+P]
    bool, tags [B,L+P,C]; negs_for(pool) -> [M,D] L2-normalised negatives of pool `pool` (-1 = the shared pool)."""
    L, P = cfg["MAX_ITEM_LIST_LENGTH"], cfg["pred_len"]
    S, C = cfg["num_segment_head"], cfg["num_prior_head"]
    B = e.shape[0]
    hi = cfg["head_interaction"]
    loss_kind = cfg["loss"]
    by_cat = bool(cfg.get("neg_sample_by_cat")) and loss_kind == "prior"
    thres = cfg.get("nce_thres") or 0.99
    lam = horizon_discount(cfg)
    names = cfg.get("int_to_category") or {c: str(c) for c in range(C)}
    heads = decode_heads(weights, cfg, out).permute(0, 2, 1, 3)          # [B,H,L,D]
    tgt, valid, tg = future_targets(e, mask, tags, L, P)
    p_grid = torch.arange(P)[None, :, None].expand(B, P, L)

    res = OrderedDict()
    res["loss"] = torch.zeros(())

    shared = None
    if (not by_cat) or (loss_kind == "prior" and hi == "additive"):
        shared = negs_for(-1 if n_pools > 1 else 0)

    if loss_kind == "nce" or (loss_kind == "prior" and hi == "additive"):
        seg_len = seg_len_of(cfg)
        head_for_p = torch.arange(P) // seg_len
        cur = heads[:, head_for_p]                                         # [B,P,L,D]
        m = valid
        logits = nce_logits(cur[m], tgt[m], shared, weights["logit_scale"], thres, operand_round=MIXED)
        tl = token_ce(logits)
        per_p = lam * _per_offset_mean(tl, p_grid[m], P)
        res["loss"] = res["loss"] + per_p.sum()
        seg = per_p.detach().view(S, seg_len).sum(1)
        for s in range(S):
            res[f"seg_{s}_loss"] = seg[s]
        i0 = p_grid[m] == 0
        if i0.any():
            res.update(train_topk_logs(logits[i0].detach()))

    if loss_kind == "prior":
        seg_len = P if hi == "additive" else seg_len_of(cfg)
        seg_for_p = torch.arange(P) // seg_len
        w_c = prior_loss_weights(cfg)
        accum = torch.zeros(P)
        for c in range(C):
            res[f"head_nce_{names[c]}_loss"] = 0
            negs = negs_for(c) if by_cat else shared
            if cfg.get("prior_switch") in ("in", "in_out") and not (cfg.get("master_switch") and c > 0):
                sw_loss, sw_acc = prior_switch_loss(weights, cfg, out, heads, tg, c, w_c)
                wsw = cfg["prior_switch_loss_weight"]
                res[f"head_cat_{names[c]}_acc"] = sw_acc
                res["loss"] = res["loss"] + wsw * sw_loss
                res[f"head_cat_{names[c]}_loss"] = wsw * sw_loss.detach()
            m = valid & tg[..., c].bool()
            if m.sum() == 0:            # hstu.py:815-839: head contributes nothing
                continue
            if hi == "additive":
                head_for_p = torch.full((P,), S + c, dtype=torch.long)
            else:
                head_for_p = seg_for_p * C + c
            cur = heads[:, head_for_p]
            logits = nce_logits(cur[m], tgt[m], negs, weights["logit_scale"], thres, operand_round=MIXED)
            tl = token_ce(logits)
            per_p = lam * w_c[c] * _per_offset_mean(tl, p_grid[m], P)
            res["loss"] = res["loss"] + per_p.sum()
            accum = accum + per_p.detach()
            res[f"head_nce_{names[c]}_loss"] = per_p.sum().detach()
            i0 = p_grid[m] == 0
            if c == 0 and i0.any():
                res.update(train_topk_logs(logits[i0].detach()))
        if hi != "additive":
            seg = accum.view(S, seg_len_of(cfg)).sum(1)
            for s in range(S):
                res[f"seg_{s}_loss"] = res.get(f"seg_{s}_loss", 0.0) + seg[s]
        else:
            res["loss"] = res["loss"] / 2
    return res


# --------------------------------------------------------------------------- #
# evaluation
# --------------------------------------------------------------------------- #
def compute_item_all(weights):
    """L2-normalised (projected) item table. model/IDNet/hstu.py:1018-1021"""
    return l2n(item_tower(weights, weights["item_embedding.weight"]))


def last_hidden(weights, cfg, item_seq):
    """Encoder on [B,L] (front zero-padded) -> output at the last position [B,D].  model/IDNet/hstu.py:879-913."""
    _, x = embed_inputs(weights, cfg, item_seq)
    return hstu_encoder(x, weights, cfg, item_seq != 0)[:, -1]


def user_head_embeddings(weights, cfg, item_seq, last=None):
    """Last position, H heads, fp32 L2-norm.  model/IDNet/hstu.py:915-931, 965-966.  Returns [B,H,D]."""
    last = last_hidden(weights, cfg, item_seq) if last is None else last
    return l2n(decode_heads(weights, cfg, last).float())


def predict_switch(weights, cfg, item_seq, last=None):
    """Prior-switch decisions at the last position: pred [B, n] bool (n = 1 with master_switch, else C).
    model/IDNet/hstu.py:935-956."""
    last = last_hidden(weights, cfg, item_seq) if last is None else last      # [B,D]
    head_out = decode_heads(weights, cfg, last)                            # [B,H,D] (not normalised)
    n = 1 if cfg.get("master_switch") else cfg["num_prior_head"]
    return torch.stack([switch_logits(weights, cfg, last, head_out, c) >= 0 for c in range(n)], dim=1)


def predict_scores(weights, cfg, item_seq, item_feature, all_item_tags, target_tags, logs=None, last=None):
    """HSTU.predict. model/IDNet/hstu.py:874-1016.

    item_feature [N,D] (re-normalised here, hstu.py:974-975), all_item_tags [C,N],
    target_tags [B,eval_pred_len,C].  Returns scores [B,H,N] with -inf masks; `logs` (optional dict) receives the
    reference's wandb counters (num_samples, head_cat_*_num_correct).
    """
    S, C = cfg["num_segment_head"], cfg["num_prior_head"]
    if last is None:            # `last` given: the caller ran its own sequence encoder (HLLM twin, hllm.py:781-783)
        last = last_hidden(weights, cfg, item_seq)
    u = user_head_embeddings(weights, cfg, item_seq, last)                 # [B,H,D]
    names = cfg.get("int_to_category") or {c: str(c) for c in range(C)}
    pred = None
    if cfg["loss"] == "prior" and cfg.get("prior_switch") in ("in", "in_out"):
        pred = predict_switch(weights, cfg, item_seq, last)
        if logs is not None:
            for c in range(pred.shape[1]):
                label = target_tags[:, :, c].sum(dim=-1) > 0
                logs[f"head_cat_{names[c]}_num_correct"] = float((label == pred[:, c]).sum())
    if logs is not None:
        logs["num_samples"] = cfg["eval_pred_len"] * item_seq.shape[0]
    it = l2n(item_feature.float())
    scores = u @ it.T                                                      # [B,H,N]
    if cfg["loss"] == "prior":
        additive = cfg["head_interaction"] == "additive"
        if cfg.get("prior_given_at_test"):
            g = cfg.get("given_prior_len") or cfg["eval_pred_len"]
            given = target_tags[:, :g].bool().any(dim=1)                   # [B,C]
            if additive:
                scores[:, S:] = scores[:, S:].masked_fill(~given[:, :, None], float("-inf"))
            else:
                scores = scores.masked_fill(~given.repeat(1, S)[:, :, None], float("-inf"))
        tagm = all_item_tags.bool()
        if additive:
            scores[:, S:] = scores[:, S:].masked_fill(~tagm[None], float("-inf"))
        else:
            scores = scores.masked_fill(~tagm.repeat(S, 1)[None], float("-inf"))
        if pred is not None and cfg.get("use_prior_switch_test"):          # hstu.py:1002-1015: heads switched off
            if cfg.get("master_switch"):
                off = torch.cat([~pred[:, :1], pred[:, :1].expand(-1, C - 1)], dim=1)          # [B,C]
            else:
                off = ~pred
            if additive:
                scores[:, S:] = scores[:, S:].masked_fill(off[:, :, None], float("-inf"))
            else:
                scores = scores.masked_fill(off.repeat(1, S)[:, :, None], float("-inf"))
    return scores
