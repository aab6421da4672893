#!/usr/bin/env python3
"""Time the HLLM twin (SURVEY a19 / BASELINE configs 3-4 shape) on one MI355X: TinyLlama-1.1B-shaped user decoder
(22 layers, d = 2048, 32 heads / 4 KV heads, ffn 5632) on frozen item embeddings, hierarchical prior heads.

  python tools/hllm_bench.py [--layers 22] [--batch 64] [--seq 50] [--steps 10] [--mode train|eval]

Prints one JSON line: ms/step, sequences/s, per-entry-point kernel time (HIP events) and the decoder's GEMM-flop rate.
Synthetic ids / tags (mhr_amd.synth), random-init weights, random item embeddings (no pretrained tower exists offline).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "multi-head-recommendation-with-human-priors_amd", "code"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--layers", type=int, default=22)
    ap.add_argument("--hidden", type=int, default=2048)
    ap.add_argument("--heads", type=int, default=32)
    ap.add_argument("--kv-heads", type=int, default=4)
    ap.add_argument("--ffn", type=int, default=5632)
    ap.add_argument("--batch", type=int, default=64)
    ap.add_argument("--seq", type=int, default=50)
    ap.add_argument("--pred", type=int, default=4)
    ap.add_argument("--cats", type=int, default=8)
    ap.add_argument("--items", type=int, default=453938)
    ap.add_argument("--negatives", type=int, default=8192)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--mode", default="train", choices=["train", "eval"])
    ap.add_argument("--profile", action="store_true", help="after the timed steps: torch.profiler over 2 steps, top device kernels on stderr")
    args = ap.parse_args()
    import torch
    import REC  # noqa: F401
    from mhr_amd import ops, synth
    from REC.config.configurator import Config, apply_run_fixups
    from REC.trainer import Trainer
    from REC.utils import get_model
    dev = torch.device("cuda", 0)
    cfgd = synth.base_config(
        model="HLLM", MAX_ITEM_LIST_LENGTH=args.seq, pred_len=args.pred, eval_pred_len=args.pred, loss="prior",
        num_prior_head=args.cats, num_segment_head=2, head_interaction="hierarchical", segment_embed=True, medusa_num_layers=1,
        eval_num_cats=args.cats, neg_sample_by_cat=True, num_negatives=args.negatives, train_batch_size=args.batch,
        eval_batch_size=args.batch, freeze_item_llm=True, dummy_llm=False, item_emb_token_n=1, topk=[5, 10, 50, 200],
        optim_args={'learning_rate': 1e-4, 'weight_decay': 0.0}, scheduler_args=None, device=dev, eval_interval=0,
        checkpoint_dir=None, save_model_note="bench",
        user_llm_config=dict(hidden_size=args.hidden, intermediate_size=args.ffn, num_hidden_layers=args.layers,
                             num_attention_heads=args.heads, num_key_value_heads=args.kv_heads, vocab_size=32000,
                             rms_norm_eps=1e-5, rope_theta=10000.0))
    cfg = apply_run_fixups(Config(config_dict=cfgd))
    data = synth.SyntheticData(cfg, args.items, dev, seed=2020)
    cfg["int_to_category"] = data.int_to_category
    torch.manual_seed(2020)
    model = get_model("HLLM")(cfg, data).to(dev)
    table = torch.randn(args.items, args.hidden, device=dev)
    model.set_all_item_embeds(table)
    trainer = Trainer(cfg)
    trainer.setup_model(model)
    B, L, P, C = args.batch, args.seq, args.pred, args.cats

    def to_interaction(b):
        items, neg, mask, tags = b
        it = dict(attention_mask=mask, pos_item_ids=items, pos_tag_categories=tags, neg_item_ids=neg[:, -1].contiguous())
        for c in range(C):
            it[f"neg_item_ids_cat{c}"] = neg[:, c].contiguous()
        return it

    if args.mode == "train":
        batches = [to_interaction(data.train_batch(B)) for _ in range(4)]
        model.train()

        def step(i):
            return trainer.train_step_fn(batches[i % len(batches)])
    else:
        batches = [data.eval_batch(B) for _ in range(4)]
        trainer.compute_item_feature(data.item_tags)

        def step(i):
            fused, pu, pi, tt, _ = trainer._full_sort_batch_eval(batches[i % len(batches)])
            trainer.eval_collector.eval_batch_collect(fused, pu, pi)

    for i in range(args.warmup):
        step(i)
    names = ["mhr_rmsnorm_fwd", "mhr_rmsnorm_bwd", "mhr_swiglu_fwd", "mhr_swiglu_bwd", "mhr_rope_inplace", "mhr_softmax_attn_fwd",
             "mhr_softmax_attn_bwd", "mhr_nce_fwd", "mhr_nce_bwd_tokens", "mhr_nce_bwd_negs", "mhr_embedding_gather_fwd",
             "mhr_catalog_score_emit_sliced", "mhr_topk_select_sliced"]
    ops.PROFILE = {k: [] for k in names}
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    last = None
    for i in range(args.steps):
        last = step(args.warmup + i)
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    prof = ops.profile_summary()
    ops.PROFILE = None
    if args.profile:
        from torch.profiler import profile, ProfilerActivity
        with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as tp:
            for i in range(2):
                step(i)
            torch.cuda.synchronize()
        rows = sorted(((getattr(e, "self_device_time_total", 0) / 2, e.count / 2, e.key) for e in tp.key_averages()), reverse=True)
        for dt, n, key in rows[:18]:
            print(f"{dt:10.1f} us/step {n:7.1f} calls  {key[:120]}", file=sys.stderr)
    D, F, nh, nkv = args.hidden, args.ffn, args.heads, args.kv_heads
    hd = D // nh
    per_tok = args.layers * 2 * (D * (nh + 2 * nkv) * hd + D * D + 3 * D * F)          # GEMM flops per token, forward
    flops = B * L * per_tok * (3 if args.mode == "train" else 1)
    out = {"workload": f"HLLM {args.mode}: user decoder {args.layers}x(d={D}, {nh}/{nkv} heads, ffn {F}), L={L} P={P} C={C} S=2 "
                       f"hierarchical, B={B}, N={args.items}, {args.negatives} negatives/pool, frozen item embeddings",
           "ms_per_step": round(1000 * el / args.steps, 3), "seq_per_s": round(B * args.steps / el, 1),
           "decoder_gemm_TFLOPs": round(flops / (el / args.steps) / 1e12, 1),
           "kernel_ms_per_step": {k: round(v[2] / args.steps, 3) for k, v in sorted(prof.items(), key=lambda kv: -kv[1][2])}}
    if last is not None:
        out["loss"] = round(float(last["loss"].detach()), 4)
    print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
