#!/usr/bin/env python3
"""Generate the golden fixtures under tests/golden/ from the REFERENCE implementation.

Runs only in the build container (needs /root/reference); the fixtures it writes
are what travels.  Usage:  PYTHONDONTWRITEBYTECODE=1 python tests/gen_golden.py

It imports the reference's own `REC/model/IDNet/hstu.py`, `REC/evaluator/*` and
`REC/utils/lr_scheduler.py` (SURVEY.md section 8c recipe: three logging-only stub
modules, a 1-rank gloo group), drives them on tiny seeded inputs and stores
inputs, weights and outputs as .npz.  No reference source text is stored.
"""
import importlib.machinery
import os
import sys
import types

sys.dont_write_bytecode = True
REF = "/root/reference/code"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _stub(name, attrs):
    m = types.ModuleType(name)
    m.__spec__ = importlib.machinery.ModuleSpec(name, None)
    for a in attrs:
        setattr(m, a, lambda *a, **k: None)
    sys.modules[name] = m


def _setup():
    if not os.path.isdir(REF):
        raise SystemExit("reference tree not present; fixtures can only be regenerated in the build container")
    sys.path.insert(0, REF)
    _stub("colorlog", ["ColoredFormatter"])
    _stub("colorama", ["init"])
    _stub("tensorboardX", ["SummaryWriter"])
    import torch.distributed as dist
    if not dist.is_initialized():
        dist.init_process_group("gloo", init_method="tcp://127.0.0.1:29533", rank=0, world_size=1)


class Cfg(dict):
    """Missing keys read as None, like REC/config/configurator.py:142-152."""

    def __getitem__(self, k):
        return dict.get(self, k, None)

    def get(self, k, default=None):
        v = dict.get(self, k, None)
        return default if v is None else v


class FakeData:
    def __init__(self, item_num, C):
        self.item_num = item_num
        self.category_to_int = {f"cat{c}": c for c in range(C)}
        self.category_counts = {f"cat{c}": 10 + 7 * c for c in range(C)}


def base_cfg(**kw):
    c = Cfg(
        item_embedding_size=16, hstu_embedding_size=16, MAX_ITEM_LIST_LENGTH=12, pred_len=1, eval_pred_len=1,
        medusa_lambda=0.99, num_segment_head=1, num_prior_head=1, head_interaction="multiplicative",
        medusa_num_layers=0, category_by="item", n_layers=2, n_heads=2, hidden_act="silu",
        hidden_dropout_prob=0.0, attn_dropout_prob=0.0, enable_relative_attention_bias=True,
        loss="nce", neg_sample_by_cat=True, pos_sample_mix_ratio=0, fix_temp=False, nce_thres=None,
        weighted_prior_loss=True, prior_switch=None, split_mode="combine",
    )
    c.update(kw)
    C = c["num_prior_head"]
    c["int_to_category"] = {i: f"cat{i}" for i in range(C)}
    return c


def cfg_json(cfg, N, C):
    import json
    fd = FakeData(N, C)
    c = dict(cfg)
    c["int_to_category"] = {str(k): v for k, v in cfg["int_to_category"].items()}
    c["category_counts"] = fd.category_counts
    c["category_to_int"] = fd.category_to_int
    c["item_num"] = N
    return json.dumps(c)


def make_batch(g, N, B, L, P, C, n_neg, pools, item_tags):
    import torch
    items = torch.randint(1, N, (B, L + P), generator=g)
    mask = torch.ones(B, L + P, dtype=torch.int64)
    for b in range(B):
        pad = int(torch.randint(0, L // 2, (1,), generator=g))
        mask[b, :pad] = 0                       # front context padding (random real ids, mask 0)
        if b % 3 == 1:
            mask[b, L + P - 1:] = 0             # short prediction window
    neg = torch.randint(1, N, (B, pools, n_neg), generator=g)
    tags = item_tags[items] if C > 1 else torch.zeros(B, 0, dtype=torch.int64)
    return items, neg, mask, tags


def np_state(model):
    return {"w/" + k: v.detach().cpu().numpy() for k, v in model.state_dict().items()}


def run_train_case(name, cfg, N, B, n_neg, seed, grads=("item_embedding.weight", "_hstu._attention_layers.0._uvqk",
                                                          "_hstu._attention_layers.0._o.weight", "logit_scale")):
    import numpy as np
    import torch
    from REC.model.IDNet.hstu import HSTU
    torch.manual_seed(seed)
    L, P, C = cfg["MAX_ITEM_LIST_LENGTH"], cfg["pred_len"], cfg["num_prior_head"]
    model = HSTU(cfg, FakeData(N, C))
    model._verbose = False
    model.train()      # dropout prob is 0 in every fixture
    g = torch.Generator().manual_seed(seed + 1)
    item_tags = (torch.rand(N, max(C, 1), generator=g) < 0.45).long()
    item_tags[torch.arange(N), torch.randint(0, max(C, 1), (N,), generator=g)] = 1
    pools = C + 1 if (cfg["loss"] == "prior" and cfg["neg_sample_by_cat"]) else 1
    batch = make_batch(g, N, B, L, P, C, n_neg, pools, item_tags)
    out = model(batch)
    loss = out["loss"]
    loss.backward()
    d = np_state(model)
    for k, t in zip(("items", "neg_items", "mask", "tags"), batch):
        d["in/" + k] = t.numpy()
    d["in/item_tags"] = item_tags.numpy()
    for k, v in out.items():
        d["out/" + k] = np.asarray(v.detach().numpy() if torch.is_tensor(v) else v, dtype=np.float64)
    named = dict(model.named_parameters())
    for k in grads:
        if k in named and named[k].grad is not None:
            d["grad/" + k] = named[k].grad.numpy()
    # every head parameter's gradient too (small)
    for k, p_ in named.items():
        if ("medusa" in k or "segment_emb" in k or "position_embedding" in k or "aux_cat_head" in k) and p_.grad is not None:
            d["grad/" + k] = p_.grad.numpy()
    d["cfg/json"] = np.asarray(cfg_json(cfg, N, C))
    np.savez_compressed(os.path.join(OUT, name + ".npz"), **d)
    print(f"{name}: loss={float(loss):.6f} keys={len(d)}")
    return model, item_tags


def run_attention_case():
    import numpy as np
    import torch
    from REC.model.IDNet import hstu as ref
    g = torch.Generator().manual_seed(7)
    d = {}
    for tag, (B, L, Hh, dq) in {"a": (3, 12, 2, 8), "b": (2, 40, 4, 16), "c": (2, 33, 1, 32)}.items():
        q = torch.randn(B, L, Hh * dq, generator=g)
        k = torch.randn(B, L, Hh * dq, generator=g)
        v = torch.randn(B, L, Hh * dq, generator=g)
        seq = torch.randint(1, 50, (B, L), generator=g)
        seq[0, : L // 3] = 0                              # padded prefix: masked keys, live query rows
        seq[-1, L // 2] = 0                               # a hole in the middle
        mask = (seq != 0).unsqueeze(1).unsqueeze(2)
        mask = torch.tril(mask.expand(-1, -1, L, -1))
        o = ref._hstu_attention_maybe_from_cache(Hh, dq, dq, q, k, v, mask)
        for n_, t in (("q", q), ("k", k), ("v", v), ("seq", seq), ("out", o)):
            d[f"{tag}/{n_}"] = t.numpy()
        d[f"{tag}/n_heads"] = np.asarray(Hh)
    np.savez_compressed(os.path.join(OUT, "attention_unit.npz"), **d)
    print("attention_unit ok")


def run_predict_case(name, cfg, N, B, seed):
    import numpy as np
    import torch
    from REC.model.IDNet.hstu import HSTU
    torch.manual_seed(seed)
    L, C = cfg["MAX_ITEM_LIST_LENGTH"], cfg["num_prior_head"]
    E = cfg["eval_pred_len"]
    model = HSTU(cfg, FakeData(N, C))
    model.eval()
    g = torch.Generator().manual_seed(seed + 5)
    item_tags = (torch.rand(N, C, generator=g) < 0.4).long()
    item_tags[torch.arange(N), torch.randint(0, C, (N,), generator=g)] = 1
    item_seq = torch.randint(1, N, (B, L), generator=g)
    for b in range(B):
        item_seq[b, : int(torch.randint(0, L - 1, (1,), generator=g))] = 0   # front zero padding
    target = torch.randint(1, N, (B, E), generator=g)
    target_tags = item_tags[target]
    feat = model.compute_item_all()
    d = np_state(model)
    d.update({"in/item_seq": item_seq.numpy(), "in/item_tags": item_tags.numpy(), "in/target_tags": target_tags.numpy(),
              "in/target": target.numpy(), "out/item_feature": feat.numpy()})
    d["cfg/json"] = np.asarray(cfg_json(cfg, N, C))
    for given in (False, True):
        model.prior_given_at_test = given
        model.given_prior_len = 2 if given else E
        scores, logs, _, _ = model.predict(item_seq, None, feat, item_tags.t().contiguous(), target_tags)
        d[f"out/scores_given{int(given)}"] = scores.numpy()
        for k_, v_ in logs.items():
            d[f"out/log_given{int(given)}/{k_}"] = np.asarray(float(v_))
    np.savez_compressed(os.path.join(OUT, name + ".npz"), **d)
    print(f"{name}: scores {tuple(scores.shape)} -inf frac {float(torch.isinf(scores).float().mean()):.3f}")


def run_collector_case(name, N, B, S, C, hi, K, E, seed, split_mode="combine", small_cat=False, topk=None):
    import numpy as np
    import torch
    from REC.evaluator.collector import Collector
    from REC.evaluator.evaluator import Evaluator
    topk = topk or [1, 5, K // 2, K]
    H = S * C if hi != "additive" else S + C
    pl = sorted({0, E - 1, max(E // 2 - 1, 0)})
    cfg = Cfg(metrics_pred_len_list=pl, eval_pred_len=E, topk=topk, head_interaction=hi, num_segment_head=S,
              num_prior_head=C, split_mode=split_mode, device="cpu", metrics=["Recall", "NDCG"],
              shared_metrics=["Entropy"], eval_num_cats=C, eval_by_cat=False, outlier_user_metrics=None,
              int_to_category={i: f"cat{i}" for i in range(C)}, metric_decimal_place=7)
    g = torch.Generator().manual_seed(seed)
    tags = (torch.rand(N, max(C, 2), generator=g) < 0.4)
    tags[torch.arange(N), torch.randint(0, max(C, 2), (N,), generator=g)] = True
    if small_cat:
        tags[:, 0] = False
        tags[torch.randperm(N, generator=g)[:5] , 0] = True       # category 0 admits only 5 items (< K)
        tags[tags.sum(1) == 0, 1] = True
    base = torch.randn(B, 1, N, generator=g)
    scores = base + 0.35 * torch.randn(B, H, N, generator=g)       # correlated heads -> cross-head duplicates
    if H > 1 and C > 1:
        tagm = tags[:, :C].t()                                     # [C,N]
        if hi == "additive":
            scores[:, S:] = scores[:, S:].masked_fill(~tagm[None], float("-inf"))
        else:
            scores = scores.masked_fill(~tagm.repeat(S, 1)[None], float("-inf"))
    hist_len = torch.randint(3, 15, (B,), generator=g)
    hist_u = torch.cat([torch.full((int(n),), b) for b, n in enumerate(hist_len)])
    hist_i = torch.randint(1, N, (int(hist_len.sum()),), generator=g)
    pos_i = torch.randint(1, N, (B, E), generator=g)
    if E > 1:
        pos_i[0, 1] = pos_i[0, 0]                                  # duplicated target
    # make some targets rank high so hits occur
    for b in range(B):
        scores[b, :, pos_i[b, 0]] += 3.0
    scores = torch.where(torch.isnan(scores), torch.full_like(scores, float("-inf")), scores)
    raw = scores.clone()
    # trainer.py:724-726
    scores[:, :, 0] = -np.inf
    scores[hist_u, :, hist_i] = -np.inf
    pos_u = torch.arange(B).unsqueeze(-1).repeat(1, E)
    col = Collector(cfg)
    col.set_all_tags(tags.long())
    det = col.eval_batch_collect(scores.clone(), pos_u, pos_i, tag_category=None, outlier_users=None,
                                 log_detailed_results=(split_mode == "combine" or H == 1))
    d = {"in/scores_raw": raw.numpy(), "in/hist_u": hist_u.numpy(), "in/hist_i": hist_i.numpy(),
         "in/positive_i": pos_i.numpy(), "in/all_tags": tags.numpy(), "cfg/K": np.asarray(K),
         "cfg/topk": np.asarray(topk), "cfg/pred_len_list": np.asarray(pl), "cfg/S": np.asarray(S),
         "cfg/C": np.asarray(C), "cfg/additive": np.asarray(hi == "additive"), "cfg/split_mode": np.asarray(split_mode)}
    if det:
        for k in ("values", "head_source", "idx", "values_by_head", "idx_by_head"):
            if k in det:
                d["out/" + k] = np.asarray(det[k])
    ev = Evaluator(cfg)
    shared = col.get_data_struct(-1)
    for k, v in ev.evaluate(shared, pred_len=-1).items():
        d["out/shared/" + k] = np.asarray(v, dtype=np.float64)
    d["out/rec_tags"] = shared.get("rec.rec_tags").numpy()
    for p in pl:
        st = col.get_data_struct(p)
        d[f"out/topk_{p}"] = st.get("rec.topk").numpy()
        for k, v in ev.evaluate(st, pred_len=p).items():
            d[f"out/m{p}/" + k] = np.asarray(v, dtype=np.float64)
    np.savez_compressed(os.path.join(OUT, name + ".npz"), **d)
    print(f"{name}: H={H} K={K} modes={split_mode} keys={len(d)}")


def run_schedule_and_adam():
    import numpy as np
    import torch
    from REC.utils.lr_scheduler import get_cosine_schedule_with_warmup
    p = torch.nn.Parameter(torch.zeros(1))
    opt = torch.optim.AdamW([p], lr=1e-3, weight_decay=0.0)
    sch = get_cosine_schedule_with_warmup(opt, num_warmup_steps=10, num_training_steps=100)
    lrs = []
    for _ in range(100):
        lrs.append(opt.param_groups[0]["lr"])
        opt.step()
        sch.step()
    # AdamW trajectory the reference's DDP branch uses (trainer.py:296-299); FusedAdam's adam_w_mode is the same maths
    g = torch.Generator().manual_seed(3)
    w = torch.nn.Parameter(torch.randn(6, 8, generator=g))
    w0 = w.detach().clone()
    opt2 = torch.optim.AdamW([w], lr=1e-2, weight_decay=0.01)
    grads, ws = [], []
    for t in range(4):
        gr = torch.randn(6, 8, generator=g)
        if t == 2:
            gr[2:] = 0                                  # untouched rows still move (dense Adam semantics)
        w.grad = gr.clone()
        opt2.step()
        grads.append(gr.numpy())
        ws.append(w.detach().clone().numpy())
    np.savez_compressed(os.path.join(OUT, "schedule_adam.npz"), lrs=np.asarray(lrs), w0=w0.numpy(),
                        grads=np.stack(grads), ws=np.stack(ws))
    print("schedule_adam ok")


def switch_cases():
    """Prior-switch fixtures (reference hstu.py:512-544, 757-805, 935-956, 1002-1015), added after the first batch:
    `python tests/gen_golden.py switch` writes only these."""
    run_train_case("hstu_switch_in", base_cfg(pred_len=4, eval_pred_len=4, num_segment_head=1, num_prior_head=3,
                                              head_interaction="multiplicative", medusa_num_layers=1, loss="prior",
                                              prior_switch="in", prior_switch_loss_weight=0.5),
                   N=401, B=5, n_neg=6, seed=41)
    run_train_case("hstu_switch_inout_asym", base_cfg(pred_len=4, eval_pred_len=4, num_segment_head=2, num_prior_head=3,
                                                      head_interaction="additive", medusa_num_layers=1, loss="prior",
                                                      prior_switch="in_out", prior_switch_loss_weight=0.25,
                                                      asym_switch_loss=True, master_switch=True, detach_aux_in=True),
                   N=401, B=5, n_neg=6, seed=42)
    run_train_case("hstu_switch_last", base_cfg(pred_len=2, eval_pred_len=2, num_segment_head=1, num_prior_head=2,
                                                head_interaction="multiplicative", medusa_num_layers=1, loss="prior",
                                                prior_switch="in_out", prior_switch_loss_weight=1.0, switch_last_only=True),
                   N=301, B=4, n_neg=5, seed=43)
    run_predict_case("predict_switch", base_cfg(pred_len=4, eval_pred_len=4, num_segment_head=1, num_prior_head=3,
                                                medusa_num_layers=1, loss="prior", prior_switch="in_out",
                                                prior_switch_loss_weight=0.5, use_prior_switch_test=True), N=257, B=8, seed=44)
    run_predict_case("predict_switch_master", base_cfg(pred_len=4, eval_pred_len=4, num_segment_head=2, num_prior_head=3,
                                                       head_interaction="additive", medusa_num_layers=1, loss="prior",
                                                       prior_switch="in", prior_switch_loss_weight=0.5,
                                                       use_prior_switch_test=True, master_switch=True), N=257, B=8, seed=45)


def run_llama_cases():
    """Decoder blocks of the HLLM twin: the reference's own REC/model/HLLM/modeling_llama.py (eager attention path) on
    tiny random-init configs -> last hidden state and parameter / input gradients.  Two shapes: grouped KV heads with
    front-padded sequences (the user decoder's call, hllm.py:501-502) and head_dim 64 without padding."""
    import numpy as np
    import torch
    from REC.model.HLLM.modeling_llama import LlamaConfig, LlamaForCausalLM
    cases = {
        "llama_decoder_gqa": dict(hidden_size=64, intermediate_size=96, num_hidden_layers=2, num_attention_heads=4,
                                  num_key_value_heads=2, B=3, L=12, pad=[5, 1, 0], seed=11, rope_theta=10000.0, eps=1e-5),
        "llama_decoder_hd64": dict(hidden_size=128, intermediate_size=160, num_hidden_layers=1, num_attention_heads=2,
                                   num_key_value_heads=2, B=2, L=40, pad=[0, 7], seed=12, rope_theta=500000.0, eps=1e-6),
    }
    for name, c in cases.items():
        torch.manual_seed(c["seed"])
        cfg = LlamaConfig(hidden_size=c["hidden_size"], intermediate_size=c["intermediate_size"],
                          num_hidden_layers=c["num_hidden_layers"], num_attention_heads=c["num_attention_heads"],
                          num_key_value_heads=c["num_key_value_heads"], vocab_size=32, max_position_embeddings=64,
                          rms_norm_eps=c["eps"], rope_theta=c["rope_theta"], pretraining_tp=1)
        cfg.use_ft_flash_attn, cfg.use_cache, cfg.output_hidden_states, cfg.return_dict = False, False, True, True
        model = LlamaForCausalLM(cfg)
        with torch.no_grad():                              # norm weights away from 1 so that their maths is visible
            for n_, p_ in model.named_parameters():
                if "norm" in n_:
                    p_.add_(0.2 * torch.randn_like(p_))
        B, L, D = c["B"], c["L"], c["hidden_size"]
        x = torch.randn(B, L, D, requires_grad=True)
        mask = torch.ones(B, L, dtype=torch.bool)
        for b, pad in enumerate(c["pad"]):
            mask[b, :pad] = False
        hidden = model(inputs_embeds=x, attention_mask=mask).hidden_states[-1]
        probe = torch.randn(B, L, D)
        loss = (hidden * probe * mask[..., None]).sum()            # padded rows are not read by any caller
        loss.backward()
        out = {"w/" + k: v.detach().numpy() for k, v in model.state_dict().items() if not k.startswith("lm_head")}
        out.update(x=x.detach().numpy(), mask=mask.numpy(), probe=probe.numpy(), hidden=hidden.detach().numpy(),
                   loss=np.float64(loss.item()), dx=x.grad.numpy())
        for k, p_ in model.named_parameters():
            keep = ("layers.0.self_attn.q_proj", "layers.0.self_attn.k_proj", "layers.0.self_attn.v_proj",
                    "layers.0.mlp.up_proj", "norm")
            if p_.grad is not None and any(t in k for t in keep):
                out["g/" + k] = p_.grad.numpy()
        import json
        out["lcfg"] = np.array(json.dumps(dict(hidden_size=D, intermediate_size=c["intermediate_size"],
                                                num_hidden_layers=c["num_hidden_layers"],
                                                num_attention_heads=c["num_attention_heads"],
                                                num_key_value_heads=c["num_key_value_heads"], rms_norm_eps=c["eps"],
                                                rope_theta=c["rope_theta"])))
        np.savez_compressed(os.path.join(OUT, name + ".npz"), **out)
        print("wrote", name, "loss", loss.item())


def run_llama_packed_case():
    """Packed `cu_input_lens` batches of the item tower (flash_self_attn.py:61-130 needs flash-attn, which is not
    importable here): packing N sequences into one row with per-sequence positions equals N independent passes of the
    reference's EAGER path (modeling_llama.py:651-682) - which does import.  The fixture holds the packed inputs and the
    concatenation of the reference's per-sequence outputs and gradients."""
    import json
    import numpy as np
    import torch
    from REC.model.HLLM.modeling_llama import LlamaConfig, LlamaForCausalLM
    c = dict(hidden_size=64, intermediate_size=96, num_hidden_layers=2, num_attention_heads=4, num_key_value_heads=2,
             rope_theta=10000.0, eps=1e-5)
    lens = [5, 17, 1, 9, 12]
    torch.manual_seed(31)
    cfg = LlamaConfig(hidden_size=c["hidden_size"], intermediate_size=c["intermediate_size"],
                      num_hidden_layers=c["num_hidden_layers"], num_attention_heads=c["num_attention_heads"],
                      num_key_value_heads=c["num_key_value_heads"], vocab_size=32, max_position_embeddings=64,
                      rms_norm_eps=c["eps"], rope_theta=c["rope_theta"], pretraining_tp=1)
    cfg.use_ft_flash_attn, cfg.use_cache, cfg.output_hidden_states, cfg.return_dict = False, False, True, True
    model = LlamaForCausalLM(cfg)
    with torch.no_grad():
        for n_, p_ in model.named_parameters():
            if "norm" in n_:
                p_.add_(0.2 * torch.randn_like(p_))
    D, T = c["hidden_size"], sum(lens)
    x = torch.randn(T, D)
    probe = torch.randn(T, D)
    hidden, dx, loss, off = [], [], 0.0, 0
    for n in lens:                                         # one eager pass per sequence: positions 0..n-1, no padding
        xi = x[off:off + n][None].clone().requires_grad_(True)
        hi = model(inputs_embeds=xi, attention_mask=torch.ones(1, n, dtype=torch.bool)).hidden_states[-1]
        li = (hi[0] * probe[off:off + n]).sum()
        li.backward()                                      # parameter gradients accumulate over the sequences
        hidden.append(hi[0].detach())
        dx.append(xi.grad[0])
        loss += li.item()
        off += n
    out = {"w/" + k: v.detach().numpy() for k, v in model.state_dict().items() if not k.startswith("lm_head")}
    out.update(x=x.numpy(), lens=np.array(lens, np.int64), probe=probe.numpy(), hidden=torch.cat(hidden).numpy(),
               loss=np.float64(loss), dx=torch.cat(dx).numpy())
    for k, p_ in model.named_parameters():
        keep = ("layers.0.self_attn.q_proj", "layers.1.self_attn.k_proj", "layers.0.self_attn.v_proj", "layers.1.mlp.down_proj", "norm")
        if p_.grad is not None and any(t in k for t in keep):
            out["g/" + k] = p_.grad.numpy()
    out["lcfg"] = np.array(json.dumps(dict(hidden_size=D, intermediate_size=c["intermediate_size"],
                                            num_hidden_layers=c["num_hidden_layers"], num_attention_heads=c["num_attention_heads"],
                                            num_key_value_heads=c["num_key_value_heads"], rms_norm_eps=c["eps"],
                                            rope_theta=c["rope_theta"])))
    np.savez_compressed(os.path.join(OUT, "llama_packed.npz"), **out)
    print("wrote llama_packed loss", loss)


def run_eval_batch_cases():
    """Eval batches of the host data path: the reference's own `REC/data/dataset/evalset.py`
    (SeqEvalDataset.__getitem__, 80-150) and `collate_fn.py` (seq_eval_collate, 59-90), loaded BY FILE PATH (the REC.data
    package pulls polars / torchvision; these two files need only torch and pytz), driven on seeded synthetic users.
    The fixture holds the raw users / tags / events and every field of every collated batch, for the item-category and the
    event-category configurations, valid and test phase."""
    import importlib.util
    import numpy as np
    import torch

    def load(path, name):
        spec = importlib.util.spec_from_file_location(name, path)
        m = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(m)
        return m
    ev = load(os.path.join(REF, "REC/data/dataset/evalset.py"), "ref_evalset")
    co = load(os.path.join(REF, "REC/data/dataset/collate_fn.py"), "ref_collate")
    g = np.random.default_rng(5)
    n_users, N, C, L, E, B = 23, 90, 3, 8, 4, 7
    tags = g.random((N, C)) < 0.45
    tags[np.arange(N), g.integers(0, C, N)] = True
    tags[:6] = [[False] * C, [True] * C, [True, False, False], [True] * C, [False, True, False], [False, False, True]]
    user_seq = [[]] + [g.integers(1, N, int(g.integers(10, 30))).tolist() for _ in range(n_users)]
    train_len = [0] + [len(s_) - 8 for s_ in user_seq[1:]]
    events = [[]] + [g.integers(0, C, len(s_)).tolist() for s_ in user_seq[1:]]
    raw_tag = g.integers(0, 6, N).tolist()

    class DL:
        pass
    dl = DL()
    dl.user_num, dl.item_num, dl.user_seq, dl.train_seq_len, dl.event_seq = len(user_seq), N, user_seq, train_len, events
    dl.item_to_info = [({} if i % 11 == 0 else {"tag_category": tags[i].tolist(), "tag": int(raw_tag[i])}) for i in range(N)]   # (a list: the "tag" outlier mode indexes it with tensors)
    dl.category_to_int = {f"cat{c}": c for c in range(C)}
    out = dict(user_flat=np.array([i for s_ in user_seq for i in s_], np.int64), user_len=np.array([len(s_) for s_ in user_seq], np.int64),
               train_len=np.array(train_len, np.int64), event_flat=np.array([e for es in events for e in es], np.int64),
               tags=np.array([([False] * C if i % 11 == 0 else tags[i].tolist()) for i in range(N)]), raw_tag=np.array(raw_tag, np.int64),
               no_info=np.array([i % 11 == 0 for i in range(N)]), L=np.int64(L), E=np.int64(E), B=np.int64(B))
    cases = {"item_cat": dict(category_by="item", outlier_user_metrics="category"),
             "item_tag": dict(category_by="item", outlier_user_metrics="tag"),
             "event": dict(category_by="event", outlier_user_metrics="event")}
    for cname, kw in cases.items():
        cfg = Cfg(eval_pred_len=E, MAX_ITEM_LIST_LENGTH=L, eval_num_cats=C, int_to_category={c: f"cat{c}" for c in range(C)},
                  timestamp_required=False, **kw)
        for phase in ("valid", "test"):
            ds = ev.SeqEvalDataset(cfg, dl, phase=phase)
            n = len(ds)
            for bi, b0 in enumerate(range(0, n, B)):
                uid, item_seq, target, (hu, hi), pos_u, _, tt, outl = co.seq_eval_collate([ds[i] for i in range(b0, min(n, b0 + B))])
                pre = f"{cname}/{phase}/{bi}/"
                out.update({pre + "uid": uid.numpy(), pre + "item_seq": item_seq.numpy(), pre + "target": target.numpy(),
                            pre + "hist_u": hu.numpy(), pre + "hist_i": hi.numpy(), pre + "pos_u": pos_u.numpy(),
                            pre + "target_tags": tt.numpy(), pre + "outlier": outl.numpy()})
    np.savez_compressed(os.path.join(OUT, "eval_batches.npz"), **out)
    print("wrote eval_batches", len(out), "arrays")


def run_baichuan_cases():
    """Baichuan2-style decoder (REC/model/HLLM/baichuan/modeling_baichuan.py: packed `W_pack` q|k|v projection, rotary base
    10000, full multi-head attention, eager softmax path) on a tiny random-init config, called the way the user decoder
    is (inputs_embeds + [B,L] attention mask): last hidden state, parameter and input gradients."""
    import json
    import numpy as np
    import torch
    from REC.model.HLLM.baichuan.configuration_baichuan import BaichuanConfig
    from REC.model.HLLM.baichuan.modeling_baichuan import BaichuanForCausalLM
    c = dict(hidden_size=64, intermediate_size=96, num_hidden_layers=2, num_attention_heads=4, B=3, L=12, pad=[5, 1, 0],
             seed=21, eps=1e-6)
    torch.manual_seed(c["seed"])
    cfg = BaichuanConfig(vocab_size=32, hidden_size=c["hidden_size"], intermediate_size=c["intermediate_size"],
                         num_hidden_layers=c["num_hidden_layers"], num_attention_heads=c["num_attention_heads"],
                         max_position_embeddings=64, rms_norm_eps=c["eps"])
    cfg.use_ft_flash_attn, cfg.use_cache, cfg.output_hidden_states, cfg.return_dict = False, False, True, True
    model = BaichuanForCausalLM(cfg)
    with torch.no_grad():
        for n_, p_ in model.named_parameters():
            if "norm" in n_:
                p_.add_(0.2 * torch.randn_like(p_))
    B, L, D = c["B"], c["L"], c["hidden_size"]
    x = torch.randn(B, L, D, requires_grad=True)
    mask = torch.ones(B, L, dtype=torch.bool)
    for b, pad in enumerate(c["pad"]):
        mask[b, :pad] = False
    hidden = model(inputs_embeds=x, attention_mask=mask).hidden_states[-1]
    probe = torch.randn(B, L, D)
    loss = (hidden * probe * mask[..., None]).sum()
    loss.backward()
    out = {"w/" + k: v.detach().numpy() for k, v in model.state_dict().items() if not k.startswith("lm_head")}
    out.update(x=x.detach().numpy(), mask=mask.numpy(), probe=probe.numpy(), hidden=hidden.detach().numpy(),
               loss=np.float64(loss.item()), dx=x.grad.numpy())
    for k, p_ in model.named_parameters():
        keep = ("layers.0.self_attn.W_pack", "layers.1.self_attn.o_proj", "layers.0.mlp.up_proj", "layers.0.mlp.down_proj", "norm")
        if p_.grad is not None and any(t in k for t in keep):
            out["g/" + k] = p_.grad.numpy()
    out["lcfg"] = np.array(json.dumps(dict(model_type="baichuan", hidden_size=D, intermediate_size=c["intermediate_size"],
                                            num_hidden_layers=c["num_hidden_layers"],
                                            num_attention_heads=c["num_attention_heads"], rms_norm_eps=c["eps"])))
    np.savez_compressed(os.path.join(OUT, "baichuan_decoder.npz"), **out)
    print("wrote baichuan_decoder loss", loss.item(), "keys", sorted(k for k in out if k.startswith("g/")))


def run_comirec_cases():
    """ComiRec baseline (REC/model/IDNet/comirec.py) on the reference itself: one training step with gradients and one
    predict call."""
    import numpy as np
    import torch
    from REC.model.IDNet.comirec import ComiRec
    N, B, L, P, n_neg = 301, 5, 12, 3, 9
    cfg = base_cfg(pred_len=P, eval_pred_len=P, interest_num=3, interest_hidden=8, loss="nce", n_layers=2, n_heads=2)
    torch.manual_seed(41)
    model = ComiRec(cfg, FakeData(N, 1))
    model._verbose = False
    model.eval()                                            # dropout off (attention_net holds one)
    with torch.no_grad():                                   # trunc-normal(0.02) leaves the interests almost tied: spread them
        for n_, p_ in model.named_parameters():
            if n_.startswith("attention_net"):
                p_.mul_(20.0)
    g = torch.Generator().manual_seed(42)
    items, neg, mask, tags = make_batch(g, N, B, L, P, 1, n_neg, 1, None)
    out = model((items, neg, mask, tags))
    out["loss"].backward()
    save = np_state(model)
    save.update({"in/items": items.numpy(), "in/neg_items": neg.numpy(), "in/mask": mask.numpy(), "cfg/json": np.array(cfg_json(cfg, N, 1))})
    for k, v in out.items():
        save["out/" + k] = np.float64(float(v))
    grads = dict(model.named_parameters())
    for k in ("attention_net.0.weight", "attention_net.3.weight", "_hstu._attention_layers.0._uvqk", "logit_scale"):
        save["grad/" + k] = grads[k].grad.numpy()
    ge = grads["item_embedding.weight"].grad
    save["grad/item_embedding.weight"] = ge.numpy()
    seq = torch.randint(1, N, (4, L), generator=g)
    seq[0, :5] = 0
    seq[2, :1] = 0
    with torch.no_grad():
        feat = model.compute_item_all()
        import contextlib, io
        with contextlib.redirect_stdout(io.StringIO()):
            scores, _, _, _ = model.predict(seq, None, feat, None, None)
    save.update({"in/item_seq": seq.numpy(), "out/item_feature": feat.numpy(), "out/scores": scores.numpy()})
    np.savez_compressed(os.path.join(OUT, "comirec_nce.npz"), **save)
    print("wrote comirec_nce loss", float(out["loss"]))


def run_remi_cases():
    """REMI (REC/model/IDNet/remi.py: ComiRec + routing regularisation + interest-aware hard negatives) on the reference
    itself: one training step with every model_out key and gradients, and one predict call."""
    import numpy as np
    import torch
    from REC.model.IDNet.remi import REMI
    N, B, L, P, n_neg = 301, 5, 12, 3, 9
    cfg = base_cfg(pred_len=P, eval_pred_len=P, interest_num=3, interest_hidden=8, loss="nce", n_layers=2, n_heads=2,
                   lambda_rr=3.0e4, beta_ihn=1.0)
    torch.manual_seed(43)
    model = REMI(cfg, FakeData(N, 1))
    model._verbose = False
    model.eval()
    with torch.no_grad():
        for n_, p_ in model.named_parameters():
            if n_.startswith("attention_net"):
                p_.mul_(120.0)       # peaky routing: the regulariser and its gradient become visible
    g = torch.Generator().manual_seed(44)
    items, neg, mask, tags = make_batch(g, N, B, L, P, 1, n_neg, 1, None)
    neg[0, -1, 0] = items[0, L]                      # a false negative: a sampled negative equal to a target
    out = model((items, neg, mask, tags))
    out["loss"].backward()
    save = np_state(model)
    save.update({"in/items": items.numpy(), "in/neg_items": neg.numpy(), "in/mask": mask.numpy(), "cfg/json": np.array(cfg_json(cfg, N, 1))})
    for k, v in out.items():
        save["out/" + k] = np.float64(float(v))
    grads = dict(model.named_parameters())
    for k in ("attention_net.0.weight", "attention_net.3.weight", "_hstu._attention_layers.0._uvqk", "logit_scale"):
        save["grad/" + k] = grads[k].grad.numpy()
    save["grad/item_embedding.weight"] = grads["item_embedding.weight"].grad.numpy()
    seq = torch.randint(1, N, (4, L), generator=g)
    seq[0, :5] = 0
    seq[2, :1] = 0
    with torch.no_grad():
        feat = model.compute_item_all()
        scores, _, _, _ = model.predict(seq, None, feat, None, None)
    save.update({"in/item_seq": seq.numpy(), "out/item_feature": feat.numpy(), "out/scores": scores.numpy()})
    np.savez_compressed(os.path.join(OUT, "remi_nce.npz"), **save)
    print("wrote remi_nce", {k: float(v) for k, v in out.items()})


def main():
    _setup()
    os.makedirs(OUT, exist_ok=True)
    if len(sys.argv) > 1 and sys.argv[1] == "switch":
        switch_cases()
        return
    if len(sys.argv) > 1 and sys.argv[1] == "baichuan":
        run_baichuan_cases()
        return
    if len(sys.argv) > 1 and sys.argv[1] == "remi":
        run_remi_cases()
        return
    if len(sys.argv) > 1 and sys.argv[1] == "llama":
        run_llama_cases()
        run_llama_packed_case()
        return
    if len(sys.argv) > 1 and sys.argv[1] == "packed":
        run_llama_packed_case()
        return
    if len(sys.argv) > 1 and sys.argv[1] == "batches":
        run_eval_batch_cases()
        return
    if len(sys.argv) > 1 and sys.argv[1] == "comirec":
        run_comirec_cases()
        return
    # F1: plain next-item NCE, identity heads
    run_train_case("hstu_nce_tiny", base_cfg(), N=501, B=4, n_neg=8, seed=11)
    # F1b: multi-step NCE with one ResBlock head per segment
    run_train_case("hstu_nce_multistep", base_cfg(pred_len=4, eval_pred_len=4, num_segment_head=2, medusa_num_layers=1),
                   N=401, B=5, n_neg=6, seed=12)
    # F2: hierarchical prior heads with segment embedding, by-category negatives, weighted loss
    run_train_case("hstu_prior_hier", base_cfg(pred_len=4, eval_pred_len=4, num_segment_head=2, num_prior_head=3,
                                               head_interaction="hierarchical", medusa_num_layers=1, loss="prior",
                                               segment_embed=True), N=401, B=5, n_neg=6, seed=13)
    # F3: multiplicative and additive prior heads
    run_train_case("hstu_prior_mult", base_cfg(pred_len=4, eval_pred_len=4, num_segment_head=2, num_prior_head=3,
                                               head_interaction="multiplicative", medusa_num_layers=1, loss="prior"),
                   N=401, B=5, n_neg=6, seed=14)
    run_train_case("hstu_prior_additive", base_cfg(pred_len=4, eval_pred_len=4, num_segment_head=2, num_prior_head=3,
                                                   head_interaction="additive", medusa_num_layers=2, loss="prior"),
                   N=401, B=5, n_neg=6, seed=15)
    # shared (not by-category) negatives, unweighted, projection tower (item dim != hstu dim), bottleneck cat head
    run_train_case("hstu_prior_proj", base_cfg(item_embedding_size=8, pred_len=2, eval_pred_len=2, num_segment_head=1,
                                               num_prior_head=2, head_interaction="hierarchical", medusa_num_layers=1,
                                               loss="prior", neg_sample_by_cat=False, weighted_prior_loss=False,
                                               head_norm=True, cat_bottleneck=True),
                   N=301, B=4, n_neg=5, seed=16, grads=("item_embedding.weight", "item_id_proj_tower.weight", "logit_scale"))
    run_attention_case()
    run_predict_case("predict_mult", base_cfg(pred_len=4, eval_pred_len=4, num_segment_head=2, num_prior_head=3,
                                              medusa_num_layers=1, loss="prior"), N=257, B=6, seed=21)
    run_predict_case("predict_additive", base_cfg(pred_len=4, eval_pred_len=4, num_segment_head=2, num_prior_head=3,
                                                  head_interaction="additive", medusa_num_layers=1, loss="prior"),
                     N=257, B=6, seed=22)
    run_collector_case("collector_combine", N=300, B=6, S=1, C=4, hi="multiplicative", K=20, E=4, seed=31)
    run_collector_case("collector_smallcat", N=300, B=5, S=2, C=3, hi="multiplicative", K=20, E=4, seed=32, small_cat=True)
    run_collector_case("collector_additive", N=300, B=4, S=2, C=3, hi="additive", K=20, E=2, seed=33)
    run_collector_case("collector_k200", N=1500, B=3, S=1, C=4, hi="multiplicative", K=200, E=8, seed=34,
                       topk=[5, 10, 20, 50, 200])
    run_collector_case("collector_single", N=300, B=5, S=1, C=1, hi="multiplicative", K=20, E=1, seed=35)
    run_collector_case("collector_average", N=300, B=5, S=1, C=4, hi="multiplicative", K=20, E=4, seed=36,
                       split_mode="average")
    run_schedule_and_adam()
    switch_cases()
    run_llama_cases()
    run_llama_packed_case()
    run_baichuan_cases()
    run_comirec_cases()
    run_remi_cases()
    run_eval_batch_cases()


if __name__ == "__main__":
    main()
