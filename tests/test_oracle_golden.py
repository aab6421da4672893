"""Pin the CPU oracle against fixtures generated from the reference (tests/gen_golden.py).

CPU only.  Tolerances: fp32 torch-CPU vs fp32 torch-CPU with different op order,
so 2e-5 relative on losses/logits/grads; integer outputs bit-exact.
"""
import json

import numpy as np
import pytest
import torch

from conftest import load_golden
from oracle import decode_oracle as DO
from oracle import hstu_oracle as HO
from oracle import metrics_oracle as MO
from oracle import optim_oracle as OO

TRAIN_CASES = ["hstu_nce_tiny", "hstu_nce_multistep", "hstu_prior_hier", "hstu_prior_mult",
               "hstu_prior_additive", "hstu_prior_proj", "hstu_switch_in", "hstu_switch_inout_asym", "hstu_switch_last"]


def cfg_of(g):
    c = json.loads(str(g["cfg/json"]))
    c["int_to_category"] = {int(k): v for k, v in c["int_to_category"].items()}
    return c


def weights_of(g, requires_grad=False):
    w = {}
    for k, v in g.items():
        if k.startswith("w/"):
            t = torch.from_numpy(np.array(v))
            if requires_grad and t.is_floating_point():
                t.requires_grad_(True)
            w[k[2:]] = t
    return HO.tie_repeated_resblocks(w)


def batch_of(g):
    return tuple(torch.from_numpy(g["in/" + k]) for k in ("items", "neg_items", "mask", "tags"))


@pytest.mark.parametrize("name", TRAIN_CASES)
def test_train_forward_and_grads(name):
    g = load_golden(name)
    cfg = cfg_of(g)
    w = weights_of(g, requires_grad=True)
    out = HO.train_forward(w, cfg, batch_of(g))
    for k, v in g.items():
        if not k.startswith("out/"):
            continue
        key = k[4:]
        assert key in out, f"oracle misses model_out key {key}"
        got = float(out[key].detach()) if torch.is_tensor(out[key]) else float(out[key])
        np.testing.assert_allclose(got, float(v), rtol=2e-5, atol=2e-6, err_msg=key)
    # no extra keys either
    assert set(out.keys()) == {k[4:] for k in g if k.startswith("out/")}
    out["loss"].backward()
    n_checked = 0
    for k, v in g.items():
        if not k.startswith("grad/"):
            continue
        got = w[k[5:]].grad
        assert got is not None, k
        scale = max(1e-6, float(np.abs(v).max()))
        np.testing.assert_allclose(got.numpy(), v, rtol=2e-4, atol=2e-5 * scale, err_msg=k)
        n_checked += 1
    assert n_checked >= 3


def test_attention_unit():
    g = load_golden("attention_unit")
    for tag in "abc":
        q, k, v = (torch.from_numpy(g[f"{tag}/{n}"]) for n in "qkv")
        seq = torch.from_numpy(g[f"{tag}/seq"])
        out = HO.hstu_attention(q, k, v, seq != 0, int(g[f"{tag}/n_heads"]))
        np.testing.assert_allclose(out.numpy(), g[f"{tag}/out"], rtol=1e-5, atol=1e-6)


@pytest.mark.parametrize("name", ["predict_mult", "predict_additive", "predict_switch", "predict_switch_master"])
def test_predict_scores(name):
    g = load_golden(name)
    cfg = cfg_of(g)
    w = weights_of(g)
    feat = HO.compute_item_all(w)
    np.testing.assert_allclose(feat.numpy(), g["out/item_feature"], rtol=1e-5, atol=1e-6)
    seq = torch.from_numpy(g["in/item_seq"])
    tags = torch.from_numpy(g["in/item_tags"]).t().contiguous()
    tt = torch.from_numpy(g["in/target_tags"])
    for given in (0, 1):
        c = dict(cfg, prior_given_at_test=bool(given), given_prior_len=2 if given else cfg["eval_pred_len"])
        logs = {}
        s = HO.predict_scores(w, c, seq, feat, tags, tt, logs=logs).numpy()
        ref = g[f"out/scores_given{given}"]
        assert np.array_equal(np.isinf(s), np.isinf(ref))
        fin = np.isfinite(ref)
        np.testing.assert_allclose(s[fin], ref[fin], rtol=1e-4, atol=2e-6)
        for k in g:                                              # the reference's wandb counters (newer fixtures only)
            if k.startswith(f"out/log_given{given}/"):
                assert float(g[k]) == logs[k.split("/", 2)[2]], k


COLLECT = ["collector_combine", "collector_smallcat", "collector_additive", "collector_k200",
           "collector_single", "collector_average"]


@pytest.mark.parametrize("name", COLLECT)
def test_collector(name):
    g = load_golden(name)
    K = int(g["cfg/K"])
    mode = str(g["cfg/split_mode"])
    scores = np.array(g["in/scores_raw"])
    DO.suppress(scores, g["in/hist_u"], g["in/hist_i"])
    H = scores.shape[1]
    pl = [int(p) for p in g["cfg/pred_len_list"]]
    if H > 1 and mode == "combine":
        v, i = DO.per_head_topk(scores, K)
        ref_v, ref_i = g["out/values_by_head"], g["out/idx_by_head"]
        fin = np.isfinite(ref_v)
        assert np.array_equal(np.isfinite(v), fin)
        assert np.array_equal(i[fin], ref_i[fin])            # bit-exact indices wherever the score is finite
        np.testing.assert_array_equal(v[fin], ref_v[fin])
        items, values, source = DO.merge_dedup(v, i, K)
        rfin = np.isfinite(g["out/values"])
        assert np.array_equal(items[rfin], g["out/idx"][rfin])
        assert np.array_equal(source[rfin], g["out/head_source"][rfin])
        # rows whose tail is -inf: only uniqueness is defined (reference order is arbitrary there)
        for b in range(items.shape[0]):
            assert len(set(items[b].tolist())) == K
    else:
        items = DO.decode_topk(scores, K, mode)
        if "out/idx" in g:
            assert np.array_equal(items, g["out/idx"])
    hits = DO.hit_matrices(items, g["in/positive_i"], pl)
    tags = g["in/all_tags"]
    full_rows = np.ones(items.shape[0], bool)
    if "out/values" in g:
        full_rows = np.isfinite(g["out/values"]).all(1)
    for p in pl:
        assert np.array_equal(hits[p][full_rows], g[f"out/topk_{p}"][full_rows])
    if full_rows.all():
        topk = [int(k) for k in g["cfg/topk"]]
        for p in pl:
            m = MO.recall_ndcg(hits[p], topk)
            for k, val in m.items():
                np.testing.assert_allclose(val, float(g[f"out/m{p}/{k}"]), rtol=1e-12, err_msg=k)
        ent = MO.entropy(tags[items].astype(np.int64), topk)
        for k, val in ent.items():
            # the reference accumulates entropy in float32 (a torch tensor fed to numpy); the oracle uses float64
            np.testing.assert_allclose(val, float(g[f"out/shared/{k}"]), rtol=1e-6, err_msg=k)


def test_schedule_and_adam():
    g = load_golden("schedule_adam")
    lrs = [1e-3 * OO.cosine_warmup_factor(s, 10, 100) for s in range(100)]
    np.testing.assert_allclose(lrs, g["lrs"], rtol=1e-12)
    w = torch.from_numpy(np.array(g["w0"]))
    m, v = torch.zeros_like(w), torch.zeros_like(w)
    for t in range(g["grads"].shape[0]):
        OO.adamw_step(w, torch.from_numpy(g["grads"][t]), m, v, t + 1, lr=1e-2, weight_decay=0.01)
        np.testing.assert_allclose(w.numpy(), g["ws"][t], rtol=2e-6, atol=1e-7)


# ------------------------------------------------------------------------------------------------
# HLLM twin: the Llama decoder restatement against the reference's own modeling_llama.py (eager path)
# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("name", ["llama_decoder_gqa", "llama_decoder_hd64", "baichuan_decoder"])
def test_llama_decoder_oracle_matches_reference(name):
    """baichuan_decoder: the same restatement behind the packed W_pack projection, against the reference's own
    REC/model/HLLM/baichuan/modeling_baichuan.py."""
    import json
    from oracle import hllm_oracle as LO
    g = load_golden(name)
    lcfg = json.loads(str(g["lcfg"]))
    w = {k[2:]: torch.tensor(v).requires_grad_(True) for k, v in g.items() if k.startswith("w/")}
    x = torch.tensor(g["x"]).requires_grad_(True)
    mask = torch.tensor(g["mask"])
    hidden = (LO.baichuan_decoder if lcfg.get("model_type") == "baichuan" else LO.llama_decoder)(w, lcfg, x, mask)
    m = mask[..., None]
    ref = torch.tensor(g["hidden"])
    assert float(((hidden - ref) * m).abs().max()) <= 1e-5 * float(ref.abs().max())       # valid positions (see LO.attention)
    loss = (hidden * torch.tensor(g["probe"]) * m).sum()
    loss.backward()
    assert abs(float(loss) - float(g["loss"])) <= 1e-5 * abs(float(g["loss"]))
    assert float((x.grad - torch.tensor(g["dx"])).abs().max()) <= 1e-5 * float(np.abs(g["dx"]).max())
    n = 0
    for k, v in g.items():
        if k.startswith("g/"):
            assert float((w[k[2:]].grad - torch.tensor(v)).abs().max()) <= 1e-5 * float(np.abs(v).max()), k
            n += 1
    assert n >= 7


def test_llama_packed_sequences_oracle_matches_reference():
    """Packed `cu_input_lens` batches (flash_self_attn.py:61-130): the oracle's segment-restricted attention on ONE packed
    row with per-sequence positions reproduces the concatenation of the reference's per-sequence eager passes
    (tests/golden/llama_packed.npz, generated by the reference's own modeling_llama.py): outputs, loss, gradients."""
    import json
    from oracle import hllm_oracle as LO
    g = load_golden("llama_packed")
    lcfg = json.loads(str(g["lcfg"]))
    lens = g["lens"].tolist()
    w = {k[2:]: torch.tensor(v).requires_grad_(True) for k, v in g.items() if k.startswith("w/")}
    x = torch.tensor(g["x"])[None].requires_grad_(True)                                   # [1, T, D]: one packed row
    seg = torch.repeat_interleave(torch.arange(len(lens)), torch.tensor(lens))[None]
    pos = torch.cat([torch.arange(n) for n in lens])[None]
    hidden = LO.llama_decoder(w, lcfg, x, None, position_ids=pos, seg_ids=seg)[0]
    ref = torch.tensor(g["hidden"])
    assert float((hidden - ref).abs().max()) <= 1e-5 * float(ref.abs().max())
    loss = (hidden * torch.tensor(g["probe"])).sum()
    loss.backward()
    assert abs(float(loss) - float(g["loss"])) <= 1e-5 * abs(float(g["loss"]))
    assert float((x.grad[0] - torch.tensor(g["dx"])).abs().max()) <= 1e-5 * float(np.abs(g["dx"]).max())
    n = 0
    for k, v in g.items():
        if k.startswith("g/"):
            assert float((w[k[2:]].grad - torch.tensor(v)).abs().max()) <= 2e-5 * float(np.abs(v).max()), k
            n += 1
    assert n >= 6


# ------------------------------------------------------------------------------------------------
# ComiRec baseline: multi-interest read-out on the HSTU encoder
# ------------------------------------------------------------------------------------------------
def test_comirec_oracle_matches_reference():
    from oracle import comirec_oracle as CO
    g = load_golden("comirec_nce")
    cfg = cfg_of(g)
    w = weights_of(g, requires_grad=True)
    batch = tuple(torch.from_numpy(g["in/" + k]) for k in ("items", "neg_items", "mask")) + (None,)
    out = CO.train_forward(w, cfg, batch)
    for k, v in g.items():
        if k.startswith("out/") and k not in ("out/item_feature", "out/scores"):
            assert abs(float(out[k[4:]]) - float(v)) <= 2e-5 * max(1.0, abs(float(v))), k
    out["loss"].backward()
    for k, v in g.items():
        if k.startswith("grad/"):
            got = w[k[5:]].grad
            assert float((got - torch.from_numpy(v)).abs().max()) <= 2e-5 * float(np.abs(v).max()) + 1e-7, k
    with torch.no_grad():
        feat = HO.compute_item_all(w)
        np.testing.assert_allclose(feat.numpy(), g["out/item_feature"], rtol=1e-5, atol=1e-6)
        scores = CO.predict_scores(w, cfg, torch.from_numpy(g["in/item_seq"]), feat)
    np.testing.assert_allclose(scores.numpy(), g["out/scores"], rtol=2e-5, atol=2e-6)


def test_remi_oracle_matches_reference():
    """REMI (ComiRec + routing regulariser + interest-aware hard negatives): the restatement against the reference's own
    remi.py - every model_out key, gradients, predict scores."""
    from oracle import remi_oracle as RO
    g = load_golden("remi_nce")
    cfg = cfg_of(g)
    w = weights_of(g, requires_grad=True)
    batch = tuple(torch.from_numpy(g["in/" + k]) for k in ("items", "neg_items", "mask")) + (None,)
    out = RO.train_forward(w, cfg, batch)
    n = 0
    for k, v in g.items():
        if k.startswith("out/") and k not in ("out/item_feature", "out/scores"):
            assert abs(float(out[k[4:]]) - float(v)) <= 2e-5 * max(1.0, abs(float(v))) + (1e-9 if "rr" in k else 0), k
            n += 1
    assert n >= 5 and float(g["out/rr_loss"]) * cfg["lambda_rr"] > 0.5                      # the regulariser is a visible part of the loss
    out["loss"].backward()
    for k, v in g.items():
        if k.startswith("grad/"):
            got = w[k[5:]].grad
            ref = torch.from_numpy(v)
            assert float((got - ref).abs().max()) <= 3e-5 * float(ref.abs().max()) + 1e-8, k
    with torch.no_grad():
        sc = RO.predict_scores({k: t.detach() for k, t in w.items()}, cfg, torch.from_numpy(g["in/item_seq"]), torch.from_numpy(g["out/item_feature"]))
    assert float((sc - torch.from_numpy(g["out/scores"])).abs().max()) <= 2e-5
