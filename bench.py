#!/usr/bin/env python3
"""Headline benchmark: user-sequences/sec for HSTU Pixel8M-shaped training (BASELINE.json metric) on N MI355X.

    python bench.py --gpus N --steps K --warmup W          (N > 1: launched by torch.distributed.run, one rank per GPU)

A "step" is one full optimisation step of the hot path on one synthetic batch per rank: item gather, 8 HSTU layers,
4 prior heads, per-category sampled softmax over 8192 negatives, backward, RCCL gradient exchange, fused AdamW over all
parameters including the 454k x 256 item table.  Workload = BASELINE.json configs[1] (cfg1: L=200, P=8, D=256,
8 layers x 8 heads, C=4 prior heads, N=453 938 items, B=128 per GPU, bf16-mixed).  Inputs are generated on the device
before the timed region.  Rank 0 prints ONE JSON line.

Extra objects on the line:
  roofline     - the dominant kernel of the step (by summed HIP-event time inside the timed region), priced by its
                 ALGORITHMIC flops against the dense bf16 MFMA peak (2.5 PFLOP/s) or, for
                 HBM-bound kernels, algorithmic bytes against 8 TB/s;
  cpu_baseline - the CPU oracle's train step on the host cores, on a bounded sample of the same workload.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "multi-head-recommendation-with-human-priors_amd", "code"))

MFMA_PEAK_TFLOPS = 2500.0     # dense bf16, MI355X_MICROARCH.md
HBM_PEAK_GBS = 8000.0


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--config", default="cfg1")
    ap.add_argument("--mode", default="train", choices=["train", "eval"])
    ap.add_argument("--batch", type=int, default=0, help="per-GPU batch override (0 = config)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-host-probe", action="store_true", help="skip the 5 untimed steps that measure the host's issue time (profiler runs)")
    ap.add_argument("--no-kernel-events", action="store_true", help="skip per-kernel HIP events (pure wall-clock run)")
    return ap.parse_args()


def cpu_baseline_train(cfgd, item_num, seconds=20.0):
    """The oracle's fp32 train step (forward + autograd backward + dense AdamW over every parameter) on the host."""
    import torch
    from oracle import hstu_oracle as HO
    from oracle import optim_oracle as OO
    import mhr_amd.synth as synth
    from REC.config.configurator import Config
    from REC.utils import get_model
    # the GPU box grants a CPU share (16 cores for one GPU) although os.cpu_count() reports the whole host:
    # oversubscribing 256 threads made one oracle step take minutes
    try:
        cores = len(os.sched_getaffinity(0))
    except AttributeError:
        cores = os.cpu_count() or 1
    cores = max(1, min(cores, 16))
    torch.set_num_threads(cores)
    cfg = Config(config_dict=dict(cfgd, device="cpu"))
    data = synth.SyntheticData(cfg, item_num, "cpu")
    cfg["int_to_category"] = data.int_to_category
    torch.manual_seed(2020)
    model = get_model("HSTU")(cfg, data)
    w = {k: v.detach().clone() for k, v in model.state_dict().items()}
    params = {k: w[k].requires_grad_(True) for k, _ in model.named_parameters()}
    w.update(params)
    w = HO.tie_repeated_resblocks(w)
    ocfg = dict(cfgd, category_counts=data.category_counts, category_to_int=data.category_to_int,
                int_to_category=data.int_to_category)
    B = 4
    ocfg["num_negatives"] = cfgd["num_negatives"]
    data.config = dict(cfgd, num_negatives=cfgd["num_negatives"])
    state = {k: (torch.zeros_like(p), torch.zeros_like(p)) for k, p in params.items()}
    n, t_used = 0, 0.0
    # negatives per sample sized as on the GPU (ceil(num_negatives / B_gpu)); the sample is B=4 windows per step
    n_neg_gpu = data.n_neg(cfgd["train_batch_size"])
    while t_used < seconds and n < 40:
        batch = data.train_batch(B)
        batch = (batch[0], batch[1][:, :, :n_neg_gpu].contiguous(), batch[2], batch[3])
        t0 = time.time()
        out = HO.train_forward(w, ocfg, batch)
        out["loss"].backward()
        with torch.no_grad():
            for k, p in params.items():
                if p.grad is None:
                    continue
                OO.adamw_step(p, p.grad, state[k][0], state[k][1], n + 1, 1e-4)
                p.grad = None
        dt = time.time() - t0
        if n > 0 or dt > seconds:          # first step warms the allocator
            t_used += dt
        n += 1
    steps = max(1, n - 1)
    return {"value": round(B * steps / max(t_used, 1e-9), 3), "unit": "seq/s", "cores": cores, "kind": "port",
            "sample": f"{steps} oracle train steps (fwd+bwd+dense AdamW, fp32) of {B} windows at the cfg1 shape with "
                      f"{n_neg_gpu * B} negatives per pool (GPU batch uses {cfgd['num_negatives']})"}


def main():
    args = parse()
    import torch
    import torch.distributed as dist
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the product path has no CPU fallback")
    # one process per GPU (the driver's launch).  MHR_DIST_BACKEND=gloo with several ranks sharing one card is only for
    # rehearsing the N > 1 code path on a single-GPU box.
    n_dev = torch.cuda.device_count()
    backend = os.environ.get("MHR_DIST_BACKEND", "nccl")
    torch.cuda.set_device(local_rank % n_dev)
    dev = torch.device("cuda", local_rank % n_dev)
    if world > 1:
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)
    import mhr_amd.synth as synth
    from mhr_amd import ops
    from REC.config.configurator import Config, apply_run_fixups
    from REC.trainer import Trainer
    from REC.utils import get_model

    spec = synth.CONFIGS[args.config]
    cfgd = dict(spec["cfg"], device=dev, total_iters=30000, eval_interval=0, checkpoint_dir=None, save_model_note="bench")
    if args.batch:
        cfgd["train_batch_size"] = cfgd["eval_batch_size"] = args.batch
    cfg = apply_run_fixups(Config(config_dict=cfgd))
    N = spec["item_num"]
    data = synth.SyntheticData(cfg, N, dev, seed=2020, rank=rank, world=world)
    cfg["int_to_category"] = data.int_to_category
    torch.manual_seed(2020)
    model = get_model("HSTU")(cfg, data).to(dev)
    trainer = Trainer(cfg)
    trainer.setup_model(model)
    trainer.train_step = 3000            # past the warm-up so lr > 0: every step really moves the parameters
    B = cfg["train_batch_size"] if args.mode == "train" else cfg["eval_batch_size"]
    L, P, D, C = cfg["MAX_ITEM_LIST_LENGTH"], cfg["pred_len"], cfg["hstu_embedding_size"], data.C

    def sync():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    n_steps = args.warmup + args.steps
    if args.mode == "train":
        batches = [data.train_batch(B) for _ in range(min(n_steps, 8))]
        model.train()

        def step(i):
            return trainer.train_step_fn(batches[i % len(batches)])
    else:
        batches = [data.eval_batch(B) for _ in range(min(n_steps, 4))]
        trainer.compute_item_feature(data.item_tags)

        def step(i):
            fused, pu, pi, tt, _ = trainer._full_sort_batch_eval(batches[i % len(batches)])
            trainer.eval_collector.eval_batch_collect(fused, pu, pi)
            return None

    for i in range(args.warmup):
        step(i)
    kernels = ["mhr_nce_fwd", "mhr_nce_bwd_tokens", "mhr_nce_bwd_negs", "mhr_nce_shared_fwd_tokens", "mhr_nce_shared_bwd_tokens", "mhr_catalog_score_emit", "mhr_catalog_score_emit_sliced",
               "mhr_hstu_attn_fwd", "mhr_hstu_attn_bwd", "mhr_adam_rows", "mhr_adam_rows_lazy", "mhr_embedding_gather_fwd", "mhr_sparse_rows_segment_sum",
               "mhr_topk_select", "mhr_topk_select_sliced"]
    if not args.no_kernel_events:
        ops.PROFILE = {k: [] for k in kernels}
    sync()
    t0 = time.perf_counter()
    last = None
    for i in range(args.steps):
        last = step(args.warmup + i)
    if args.mode == "train" and getattr(trainer.optimizer, "lazy", False):
        # lazy table optimizer: rows without gradient are replayed when next read.  Whatever is still deferred at the end of the
        # timed region is paid INSIDE it (one flush of the whole table), so no optimizer work of these K steps is left out.
        trainer.optimizer.flush_table()
    host_enqueue = time.perf_counter() - t0          # the host is done issuing; the GPU may still be working
    sync()
    elapsed = time.perf_counter() - t0
    host_alone = []                                   # host cost of issuing ONE step into an idle queue (outside the timed region)
    prof_keep, ops.PROFILE = ops.PROFILE, None        # (and outside the per-kernel event collection)
    for i in range(0 if args.no_host_probe else 5):
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        step(args.warmup + args.steps + i)
        host_alone.append(time.perf_counter() - t1)
    sync()
    host_alone = sorted(host_alone)[len(host_alone) // 2] if host_alone else float('nan')
    ops.PROFILE = prof_keep
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t)
    prof_max = {}
    if ops.PROFILE is not None:           # median of each step's longest launch (robust against one slow outlier)
        for k_ in list(ops.PROFILE):
            raw = sorted(ops.profile_raw(k_), reverse=True)[:args.steps]
            if raw:
                prof_max[k_] = raw[len(raw) // 2]
    prof = ops.profile_summary() if ops.PROFILE is not None else {}
    ops.PROFILE = None

    if rank == 0:
        value = world * B * args.steps / elapsed
        out = {
            "metric": "user-sequences/sec/node (HSTU Pixel8M, seqlen 200)" if args.mode == "train" else "eval users/sec/node (HSTU Pixel8M full-catalog multi-head decode)",
            "value": round(value, 2), "unit": "seq/s" if args.mode == "train" else "users/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(1000 * elapsed / args.steps, 3), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
            "config": {"workload": f"{args.config}: HSTU {'Pixel8M' if args.config != 'cfg2' else 'MerRec'}-shaped {args.mode} step, L={L} P={P} D={D} {cfg['n_layers']} layers x "
                                   f"{cfg['n_heads']} heads, {C} prior heads, N={N} items, {cfg['num_negatives']} negatives/pool, "
                                   f"B={B}/GPU, loss={cfg['loss']}, bf16-mixed, fused AdamW over all parameters"
                                   + (" (item table: lazy replay of gradient-free steps, flushed inside the timed region)"
                                      if getattr(trainer.optimizer, "lazy", False) and args.mode == "train" else ""),
                       "global_batch": world * B, "seq_len": L, "parallelism": f"dp{world}"},
        }
        out["host_enqueue_ms_per_step"] = round(1000 * host_enqueue / args.steps, 3)
        if host_alone == host_alone:
            out["host_issue_ms_idle_queue"] = round(1000 * host_alone, 3)
        if last is not None and args.mode == "train":
            out["loss"] = round(float(last["loss"].detach()), 4)
        # ---- roofline of the dominant kernel ----
        if prof:
            dom = max(prof.items(), key=lambda kv: kv[1][2])
            if args.mode == "eval":      # the eval roofline is quoted on the catalog scorer (north star), whatever else is slow
                for cand_name in ("mhr_catalog_score_emit_sliced", "mhr_catalog_score_emit"):
                    if cand_name in prof and prof[cand_name][0] > 0:
                        dom = (cand_name, prof[cand_name])
                        break
            per_step = {k: round(v[2] / args.steps, 3) for k, v in sorted(prof.items(), key=lambda kv: -kv[1][2])}
            # the dominant kernel of the step; if it is one of the small glue kernels without a roofline model, the next one
            cands = [dom] + [kv for kv in sorted(prof.items(), key=lambda kv: -kv[1][2]) if kv[0] != dom[0]]
            for name, (launches, mean_ms, total_ms) in cands:
                if "roofline" in out:
                    break
                if name in ("mhr_nce_fwd", "mhr_nce_bwd_negs"):
                    # algorithmic flops per launch: ONE launch serves all groups (prior categories): sum over groups of
                    # live tokens x negatives x D; N_tok measured on the last batch of the timed region
                    items, _, mask, tags = batches[(args.warmup + args.steps - 1) % len(batches)]
                    idx = torch.arange(L, device=dev)[None, :] + 1 + torch.arange(P, device=dev)[:, None]
                    mb = mask.bool()
                    valid = mb[:, None, :L] & mb[:, idx]
                    if cfg["loss"] == "prior":
                        n_tok = float((valid[..., None] & tags[:, idx].bool()).sum())          # summed over the C groups
                        n_groups = C
                    else:
                        n_tok = float(valid.sum())
                        n_groups = 1
                    n_neg = world * B * data.n_neg(B)
                    # fused forward: neg logits + false-negative logits + the token-side gradient product = 3 x 2 N_tok N_neg D
                    # ALGORITHMIC flop (what the reference's formulation computes per token; the false-negative
                    # product is executed once per target row instead of once per token); negative-side backward: dN only (its
                    # logit recompute is not counted)
                    mult = 6.0 if name == "mhr_nce_fwd" else 2.0
                    flops = mult * n_tok * n_neg * D
                    ach = flops / (mean_ms * 1e-3) / 1e12
                    out["roofline"] = {"bound": "mfma", "kernel": name, "achieved": round(ach, 2), "peak": MFMA_PEAK_TFLOPS,
                                       "unit": "TFLOP/s", "frac": round(ach / MFMA_PEAK_TFLOPS, 4), "traffic": None,
                                       "launch_ms": round(mean_ms, 4), "launches_per_step": launches / args.steps,
                                       "algorithmic_flops_per_launch": flops, "tokens_per_launch": n_tok, "groups": n_groups,
                                       "negatives_per_group": n_neg}
                elif name in ("mhr_hstu_attn_fwd", "mhr_hstu_attn_bwd"):
                    # one launch = one layer over the B sequences of the batch.  Algorithmic work: forward 4 L^2 D flop
                    # per (sequence, layer) - the two products over the full L x L score matrix, half of it causally useful -
                    # backward twice that (dV, dP, dQ, dK); algorithmic bytes 4 L D 2 B forward (q, k, v in, out), 7 L D 2 B backward
                    bwd = name.endswith("bwd")
                    flops = (8.0 if bwd else 4.0) * B * L * L * D
                    nbytes = (7.0 if bwd else 4.0) * B * L * D * 2.0
                    ach = flops / (mean_ms * 1e-3) / 1e12
                    out["roofline"] = {"bound": "mfma", "kernel": name, "achieved": round(ach, 2), "peak": MFMA_PEAK_TFLOPS,
                                       "unit": "TFLOP/s", "frac": round(ach / MFMA_PEAK_TFLOPS, 4), "traffic": None,
                                       "launch_ms": round(mean_ms, 4), "launches_per_step": launches / args.steps,
                                       "algorithmic_flops_per_launch": flops, "algorithmic_bytes_per_launch": nbytes,
                                       "algorithmic_GBps": round(nbytes / (mean_ms * 1e-3) / 1e9, 1),
                                       "note": "L = 200, head_dim = 32: 7 x 7 tiles of 32 x 32 per (sequence, head); bound by the gate "
                                               "arithmetic (sigmoid + derivative per score) and the per-workgroup prologue, not by MFMA or HBM"}
                elif name in ("mhr_catalog_score_emit", "mhr_catalog_score_emit_sliced"):
                    # the launch that scores the whole catalog (the two threshold-sample launches of a step score 1/14 + 1/222 of
                    # it and are latency-sized): algorithmic flops of that launch / its own duration (the longest of the step)
                    flops = 2.0 * B * model.medusa_num_heads * D * N
                    full_ms = prof_max.get(name, mean_ms)
                    ach = flops / (full_ms * 1e-3) / 1e12
                    out["roofline"] = {"bound": "mfma", "kernel": name, "achieved": round(ach, 2), "peak": MFMA_PEAK_TFLOPS,
                                       "unit": "TFLOP/s", "frac": round(ach / MFMA_PEAK_TFLOPS, 4), "traffic": None,
                                       "launch_ms": round(full_ms, 4), "launches_per_step": launches / args.steps,
                                       "algorithmic_flops_per_launch": flops,
                                       "all_launches_ms_per_step": round(mean_ms * launches / args.steps, 4)}
                else:
                    nbytes = {"mhr_adam_rows": N * D * 24.0, "mhr_embedding_gather_fwd": None}.get(name)
                    if nbytes:
                        ach = nbytes / (mean_ms * 1e-3) / 1e9
                        out["roofline"] = {"bound": "hbm", "kernel": name, "achieved": round(ach, 1), "peak": HBM_PEAK_GBS,
                                           "unit": "GB/s", "frac": round(ach / HBM_PEAK_GBS, 4), "traffic": None,
                                           "launch_ms": round(mean_ms, 4)}
            # HBM bytes per launch of that kernel from the PMC passes (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE run on this same
            # command and condensed by tools/summarize_profiles.py; counters cannot be read from inside the process)
            name = out["roofline"]["kernel"] if "roofline" in out else name
            if "roofline" in out and args.config == "cfg1" and not args.batch:
                # one C-ABI call may be several launches (mhr_nce_fwd = false-negative bit table + fused forward): their sum
                knames = {"mhr_nce_bwd_tokens": ["nce_bwd_rows_kernel"], "mhr_nce_bwd_negs": ["nce_bwd_n_kernel"],
                          "mhr_hstu_attn_bwd": ["hstu_attn_bwd_kernel"], "mhr_hstu_attn_fwd": ["hstu_attn_fwd_kernel"],
                          "mhr_nce_fwd": ["nce_fwd_d_kernel", "nce_fix_bits_kernel"],
                          "mhr_catalog_score_emit": ["catalog_emit_kernel"], "mhr_catalog_score_emit_sliced": ["catalog_emit_sliced_kernel"],
                          "mhr_adam_rows": ["adam_rows_kernel"]}.get(name)
                import glob
                files = sorted(glob.glob(os.path.join(ROOT, "profiles", f"r*_hbm_traffic{'_eval' if args.mode == 'eval' else ''}.json")))
                if knames and files:
                    recs = [json.load(open(files[-1]))["kernels"].get(k) for k in knames]
                    if recs[0]:
                        key_ = "hbm_bytes_max_launch" if (args.mode == "eval" and "hbm_bytes_max_launch" in recs[0]) else "hbm_bytes_per_launch"
                        out["roofline"]["traffic"] = sum(r[key_] for r in recs if r)
                        out["roofline"]["traffic_source"] = os.path.relpath(files[-1], ROOT)
            out["kernel_ms_per_step"] = per_step
            if args.mode == "train" and "mhr_nce_fwd" in prof and cfg["loss"] == "prior":
                # sampled softmax: ALGORITHMIC = the reference's per-token formulation (3 products forward, dN backward);
                # EXECUTED = what the row-sharing kernels run (2 products per distinct query row each way)
                items_, _, mask_, tags_ = batches[(args.warmup + args.steps - 1) % len(batches)]
                idx_ = torch.arange(L, device=dev)[None, :] + 1 + torch.arange(P, device=dev)[:, None]
                mb_ = mask_.bool()
                live_ = (mb_[:, None, :L] & mb_[:, idx_])[..., None] & tags_[:, idx_].bool()                 # [B, P, L, C]
                n_tok_, n_row_ = float(live_.sum()), float(live_.any(dim=1).sum())
                n_neg_ = world * B * data.n_neg(B)
                shared = os.environ.get("MHR_NCE_SHARE_ROWS", "1") != "0" and P > 1
                units = n_row_ if shared else n_tok_
                out["sampled_softmax"] = {
                    "tokens": n_tok_, "distinct_query_rows": n_row_, "row_sharing": shared,
                    "fwd_ms": round(prof["mhr_nce_fwd"][1], 4), "bwd_negs_ms": round(prof["mhr_nce_bwd_negs"][1], 4),
                    "fwd_algorithmic_TFLOPs": round(6.0 * n_tok_ * n_neg_ * D / (prof["mhr_nce_fwd"][1] * 1e-3) / 1e12, 1),
                    "fwd_executed_TFLOPs": round(4.0 * units * n_neg_ * D / (prof["mhr_nce_fwd"][1] * 1e-3) / 1e12, 1),
                    "bwd_negs_algorithmic_TFLOPs": round(2.0 * n_tok_ * n_neg_ * D / (prof["mhr_nce_bwd_negs"][1] * 1e-3) / 1e12, 1),
                    "bwd_negs_executed_TFLOPs": round(4.0 * units * n_neg_ * D / (prof["mhr_nce_bwd_negs"][1] * 1e-3) / 1e12, 1),
                    "mfma_peak_TFLOPs": MFMA_PEAK_TFLOPS}
            if "mhr_embedding_gather_fwd" in prof and args.mode == "train":
                # item ids + negative-pool ids: table rows read and written as fp32, plus the position-added encoder input
                n_pools = C if (cfg["loss"] == "prior" and cfg.get("neg_sample_by_cat")) else 1
                n_ids = B * (L + P) + data.n_neg(B) * B * n_pools
                gbytes = (n_ids * 2 + B * L) * D * 4.0
                out["embedding_gather_GBps"] = round(gbytes / (prof["mhr_embedding_gather_fwd"][2] / args.steps * 1e-3) / 1e9, 1)
            if "mhr_adam_rows" in prof:
                ms = prof["mhr_adam_rows"][1]
                out["adam_rows_GBps"] = round(N * D * 24.0 / (ms * 1e-3) / 1e9, 1)
        if not args.no_cpu_baseline and world == 1 and args.mode == "train":
            try:
                out["cpu_baseline"] = cpu_baseline_train(dict(spec["cfg"]), N)
            except Exception as e:  # noqa: BLE001 - the baseline must not kill the bench line
                out["cpu_baseline"] = {"value": None, "error": repr(e)[:200]}
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
