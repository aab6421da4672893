#!/usr/bin/env python3
"""Reference vs CPU oracle on identical inputs: speed ratio + loss / decode agreement (BUILD CONTAINER ONLY).

BASELINE.md section 3.1: the reference's source never travels to the GPU box, so `bench.py`'s `cpu_baseline` there is the
oracle ("kind": "port").  This script ties the port's speed to the reference's own code: it imports the reference's
`REC/model/IDNet/hstu.py` and `REC/evaluator/collector.py` (tests/gen_golden.py's recipe: three logging stubs, a 1-rank
gloo group), builds the SAME model (one state_dict), feeds the SAME synthetic batches and times, on this container's cores,

  train : model(batch) -> loss.backward() -> torch AdamW over every parameter   (reference trainer.py:494-536, 292-299)
          vs oracle.hstu_oracle.train_forward -> backward -> oracle.optim_oracle.adamw_step
  eval  : predict -> scores[:, :, 0] = -inf / history -> Collector.eval_batch_collect   (trainer.py:709-726, 985-990)
          vs oracle user heads -> scores -> masks -> per-head top-k -> merge (what bench.py's cpu_baseline eval leg times)

at BASELINE.json configs[0] (L = 50, D = 64, 2 layers, 1 head, nce, N = 20 001, 8192 negatives, B = 64 / 256) and a
reduced configs[1] (L = 200, P = 8, D = 256, 8 layers x 8 heads, 4 prior heads, N = 453 938, B = 16).  >= 3 warm-up and
>= 10 timed steps for cfg0 (median); the reduced cfg1 gets 1 + 3.  Writes profiles/<round>_ref_vs_oracle_cpu.json.

    PYTHONDONTWRITEBYTECODE=1 python tools/ref_vs_oracle_cpu.py [--round r03] [--quick]
"""
import argparse
import json
import os
import statistics
import sys
import time

sys.dont_write_bytecode = True
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--round", default="r03")
    ap.add_argument("--quick", action="store_true", help="cfg0 only, fewer steps")
    args = ap.parse_args()
    import gen_golden as GG
    GG._setup()                                                       # reference on sys.path, stubs, 1-rank gloo
    import numpy as np
    import torch
    from REC.model.IDNet.hstu import HSTU as RefHSTU                  # THE REFERENCE (sys.path[0] = /root/reference/code)
    from REC.evaluator.collector import Collector as RefCollector
    from oracle import decode_oracle as DO
    from oracle import hstu_oracle as HO
    from oracle import optim_oracle as OO
    import mhr_amd  # noqa: F401
    import mhr_amd.synth as synth

    cores = len(os.sched_getaffinity(0))

    def ref_cfg(cfgd, C):
        c = GG.base_cfg(**{k: v for k, v in cfgd.items() if k not in ("int_to_category",)})
        c["int_to_category"] = {i: f"cat{i}" for i in range(C)}
        return c

    def case(name, B, Be, warm, timed, threads_list):
        spec = synth.CONFIGS[name]
        N = spec["item_num"]
        cfgd = dict(spec["cfg"], hidden_dropout_prob=0.0, attn_dropout_prob=0.0, device="cpu")
        C = cfgd["num_prior_head"] if cfgd["loss"] == "prior" else 1
        data = synth.SyntheticData(synth_cfg(cfgd), N, "cpu", seed=2020)
        rc = ref_cfg(cfgd, C)
        rc["metrics_pred_len_list"] = [0]
        rc["device"] = "cpu"

        class FD:
            item_num, category_counts, category_to_int = N, data.category_counts, data.category_to_int
        torch.manual_seed(2020)
        import contextlib
        import io
        with contextlib.redirect_stdout(io.StringIO()):                # (the reference prints one line per parameter it initialises)
            ref = RefHSTU(rc, FD())
        ref._verbose = False
        ref.train()
        opt = torch.optim.AdamW(ref.parameters(), lr=1e-4, weight_decay=0.0)
        w = {k: v.detach().clone() for k, v in ref.state_dict().items()}
        params = {k: w[k].requires_grad_(True) for k, _ in ref.named_parameters()}
        w.update(params)
        w = HO.tie_repeated_resblocks(w)
        ocfg = dict(cfgd, category_counts=data.category_counts, category_to_int=data.category_to_int,
                    int_to_category=data.int_to_category, item_num=N)
        state = {k: (torch.zeros_like(p), torch.zeros_like(p)) for k, p in params.items()}
        step_no = [0]
        data.config = dict(cfgd, num_negatives=-(-cfgd["num_negatives"] // B) * B)     # the pool keeps the config's size
        batches = [data.train_batch(B) for _ in range(4)]
        ebatches = [data.eval_batch(Be) for _ in range(2)]
        losses = {"ref": [], "oracle": []}

        def ref_train(i):
            t0 = time.perf_counter()
            out = ref(batches[i % 4])
            out["loss"].backward()
            opt.step()
            opt.zero_grad(set_to_none=True)
            dt = time.perf_counter() - t0
            losses["ref"].append(float(out["loss"]))
            return dt

        def ora_train(i):
            t0 = time.perf_counter()
            out = HO.train_forward(w, ocfg, batches[i % 4])
            out["loss"].backward()
            step_no[0] += 1
            with torch.no_grad():
                for k, p in params.items():
                    if p.grad is None:
                        continue
                    OO.adamw_step(p, p.grad, state[k][0], state[k][1], step_no[0], 1e-4)
                    p.grad = None
            dt = time.perf_counter() - t0
            losses["oracle"].append(float(out["loss"]))
            return dt

        K = 200
        tags_cn = data.item_tags.long().t().contiguous()
        col = RefCollector(GG.Cfg(metrics_pred_len_list=[0], eval_pred_len=cfgd["eval_pred_len"], topk=[5, 10, 20, 50, K],
                                  head_interaction=cfgd["head_interaction"], num_segment_head=cfgd["num_segment_head"],
                                  num_prior_head=cfgd["num_prior_head"], split_mode="combine", device="cpu",
                                  metrics=["Recall", "NDCG"], shared_metrics=["Entropy"], eval_num_cats=C, eval_by_cat=False,
                                  outlier_user_metrics=None, int_to_category={i: f"cat{i}" for i in range(C)},
                                  metric_decimal_place=7))
        col.set_all_tags(data.item_tags.long())                            # (trainer.py: compute_item_feature hands the tags over)
        merged, detail = {}, [False]

        def ref_eval(i):
            eb = ebatches[i % 2]
            t0 = time.perf_counter()
            with torch.no_grad():
                ref.eval()
                feat = ref.compute_item_all() if i < 0 else ref_feat[0]
                scores, _, _, _ = ref.predict(eb[1], None, feat, tags_cn, eb[6])
                scores[:, :, 0] = -np.inf                                         # trainer.py:724
                scores[eb[3][0], :, eb[3][1]] = -np.inf                           # trainer.py:725-726
                res = col.eval_batch_collect(scores, eb[4], eb[2], log_detailed_results=detail[0])
            dt = time.perf_counter() - t0
            if detail[0]:
                merged["ref"] = np.asarray(res["idx"])
            for p_ in col.metrics_pred_len_list:
                col.get_data_struct(p_)
            return dt

        def ora_eval(i):
            eb = ebatches[i % 2]
            t0 = time.perf_counter()
            with torch.no_grad():
                wd = {k: v.detach() for k, v in w.items()}
                u = HO.user_head_embeddings(wd, ocfg, eb[1])
                sc = (u @ ora_feat[0].T).numpy()
            if cfgd["loss"] == "prior":
                tagm = data.item_tags.t().numpy()
                for h in range(sc.shape[1]):
                    sc[:, h, ~tagm[h % C]] = -np.inf
            DO.suppress(sc, eb[3][0].numpy(), eb[3][1].numpy())
            merged["oracle"] = DO.decode_topk(sc, K, "combine")
            return time.perf_counter() - t0

        def med(fn, threads, warm_, n):
            torch.set_num_threads(threads)
            for i in range(warm_):
                fn(i)
            ts = [fn(warm_ + i) for i in range(n)]
            return statistics.median(ts), len(ts)

        out = {"workload": f"{name}: L={cfgd['MAX_ITEM_LIST_LENGTH']} P={cfgd['pred_len']} D={cfgd['hstu_embedding_size']} "
                           f"{cfgd['n_layers']} layers x {cfgd['n_heads']} heads, loss={cfgd['loss']}, N={N}, "
                           f"{cfgd['num_negatives']} negatives/pool, train B={B}, eval B={Be}, fp32, dropout 0", "train": {}, "eval": {}}
        for thr in threads_list:
            # interleave reference and oracle steps on the SAME weights trajectory (both start from one state_dict and take the
            # same AdamW steps, so their losses must agree step by step)
            tr, n1 = med(ref_train, thr, warm, timed)
            to, _ = med(ora_train, thr, warm, timed)
            out["train"][f"{thr}_threads"] = {"reference_s_per_step": round(tr, 4), "oracle_s_per_step": round(to, 4),
                                              "reference_seq_per_s": round(B / tr, 2), "oracle_seq_per_s": round(B / to, 2),
                                              "oracle_over_reference_speed": round(tr / to, 3), "timed_steps": n1, "warmup": warm}
        n_cmp = min(len(losses["ref"]), len(losses["oracle"]))
        rel = max(abs(a - b) / abs(a) for a, b in zip(losses["ref"][:n_cmp], losses["oracle"][:n_cmp]))
        out["train"]["loss_first"] = [losses["ref"][0], losses["oracle"][0]]
        out["train"]["max_rel_loss_diff_over_trajectory"] = rel
        assert rel < 5e-4, ("reference and oracle trajectories diverge", rel)
        with torch.no_grad():
            ref.eval()
            ref_feat = [ref.compute_item_all()]
            ora_feat = [HO.compute_item_all({k: v.detach() for k, v in w.items()})]
        for thr in threads_list:
            er, n2 = med(ref_eval, thr, 1 if args.quick else min(3, warm), max(3, timed // 2))
            eo, _ = med(ora_eval, thr, 1 if args.quick else min(3, warm), max(3, timed // 2))
            out["eval"][f"{thr}_threads"] = {"reference_s_per_batch": round(er, 4), "oracle_s_per_batch": round(eo, 4),
                                             "reference_users_per_s": round(Be / er, 1), "oracle_users_per_s": round(Be / eo, 1),
                                             "oracle_over_reference_speed": round(er / eo, 3), "timed_batches": n2}
        detail[0] = True                                                  # one untimed batch with the reference's detailed log: its merged ids
        ref_eval(0)
        ora_eval(0)
        same = float((merged["ref"] == merged["oracle"]).mean())
        out["eval"]["merged_topk_index_agreement"] = same        # (the two models took the same steps to ~1e-7: near-ties may swap)
        return out

    # `REC` on sys.path is the REFERENCE's package here; this repo's Config class (the dict the synthetic generator reads) is
    # loaded by file path
    import importlib.util
    spec_ = importlib.util.spec_from_file_location(
        "mhr_configurator", os.path.join(ROOT, "multi-head-recommendation-with-human-priors_amd", "code", "REC", "config", "configurator.py"))
    mod_ = importlib.util.module_from_spec(spec_)
    spec_.loader.exec_module(mod_)

    def synth_cfg(cfgd):
        return mod_.Config(config_dict=cfgd)

    res = {"host_cores": cores, "torch": torch.__version__,
           "what": "reference (imported from /root/reference/code, CPU fp32) vs this repo's oracle/ on identical weights and batches"}
    threads = [1, cores]
    res["cfg0"] = case("cfg0", 64, 256, 1 if args.quick else 3, 3 if args.quick else 10, threads)
    print(json.dumps(res["cfg0"], indent=1), flush=True)
    if not args.quick:
        res["cfg1_reduced"] = case("cfg1", 16, 16, 1, 3, [cores])
        print(json.dumps(res["cfg1_reduced"], indent=1), flush=True)
    path = os.path.join(ROOT, "profiles", f"{args.round}_ref_vs_oracle_cpu.json")
    with open(path, "w") as f:
        json.dump(res, f, indent=1)
    print("wrote", path)


if __name__ == "__main__":
    main()
