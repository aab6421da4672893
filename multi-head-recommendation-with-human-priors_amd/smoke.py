"""One small invocation of the hot path on cuda:0, checked against the CPU oracle (used by __graft_entry__.smoke)."""
import os
import sys

import torch


def run():
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = os.path.join(os.path.dirname(os.path.abspath(__file__)), "code")
    for p in (root, code):
        if p not in sys.path:
            sys.path.insert(0, p)
    if not torch.cuda.is_available():
        raise RuntimeError("smoke() needs cuda:0 (MI355X); the product path has no CPU fallback")
    import mhr_amd.synth as synth
    from oracle import decode_oracle as DO
    from oracle import hstu_oracle as HO
    from REC.config.configurator import Config, apply_run_fixups
    from REC.evaluator import Collector
    from REC.utils import get_model
    from mhr_amd.optim import FusedAdamW

    dev = torch.device("cuda", 0)
    torch.manual_seed(2020)
    cfgd = synth.base_config(MAX_ITEM_LIST_LENGTH=24, pred_len=4, eval_pred_len=4, n_layers=2, n_heads=2, item_embedding_size=32,
                             hstu_embedding_size=32, loss='prior', medusa_num_layers=1, num_prior_head=3, num_segment_head=2,
                             head_interaction='multiplicative', eval_num_cats=3, num_negatives=96, hidden_dropout_prob=0.0,
                             topk=[5, 10, 20], device=dev)
    cfg = apply_run_fixups(Config(config_dict=cfgd))
    N, B = 1501, 6
    data = synth.SyntheticData(cfg, N, dev)
    cfg["int_to_category"] = data.int_to_category
    model = get_model("HSTU")(cfg, data).to(dev)
    batch = data.train_batch(B)
    # ---- one training step: forward + backward + fused Adam, checked against the fp32 oracle ----
    weights = {k: v.detach().float().cpu().clone() for k, v in model.state_dict().items()}
    out = model(batch)
    out["loss"].backward()
    ocfg = dict(cfgd, category_counts=data.category_counts, category_to_int=data.category_to_int, int_to_category=data.int_to_category)
    ref = HO.train_forward(HO.tie_repeated_resblocks(weights), ocfg, tuple(t.cpu() for t in batch))
    got, want = float(out["loss"]), float(ref["loss"])
    assert abs(got - want) < 3e-2 * abs(want), f"train loss {got} vs oracle {want}"
    opt = FusedAdamW(model, lr=1e-3)
    opt.step()
    opt.zero_grad()
    # ---- one eval batch: fused catalog scoring + top-k + merge, indices checked against the oracle decode ----
    model.eval()
    eb = data.eval_batch(4)
    feat = model.compute_item_all()
    tags = data.item_tags.long().t().contiguous()
    fused = model.predict_topk(eb[1], feat, tags, eb[6], eb[3], k=20)
    col = Collector(cfg)
    col.eval_batch_collect(fused, eb[4], eb[2])
    w2 = {k: v.detach().float().cpu().clone() for k, v in model.state_dict().items()}
    # oracle on the kernel's operand precision (bf16-rounded users and items)
    u = HO.user_head_embeddings(HO.tie_repeated_resblocks(w2), ocfg, eb[1].cpu())
    ub = u.to(torch.bfloat16).float()
    ib = HO.l2n(feat.float().cpu()).to(torch.bfloat16).float()
    sc = (ub @ ib.T).numpy().copy()
    tagm = tags.bool().cpu().numpy()
    S, C = cfg["num_segment_head"], cfg["num_prior_head"]
    import numpy as np
    for h in range(S * C):
        sc[:, h, ~tagm[h % C]] = -np.inf
    DO.suppress(sc, eb[3][0].cpu().numpy(), eb[3][1].cpu().numpy())
    rv, ri = DO.per_head_topk(sc, 20)
    gv = fused.values.cpu().numpy()
    # encoder activations differ at bf16 level between the GPU path and the fp32 oracle, so compare the overlap
    overlap = np.mean([len(set(ri[b, h]) & set(fused.indices[b, h].tolist())) / 20 for b in range(4) for h in range(S * C)])
    assert overlap > 0.8, f"top-k overlap with the fp32 oracle {overlap}"
    print(f"smoke ok: train loss {got:.5f} (oracle {want:.5f}), eval top-20 overlap with fp32 oracle {overlap:.3f}")
