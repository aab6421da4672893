#!/usr/bin/env python3
"""Headline benchmark: user-sequences/sec for HSTU Pixel8M-shaped training (BASELINE.json metric) on N MI355X.

    python bench.py --gpus N --steps K --warmup W

N > 1 without a torch.distributed launcher around it: this process starts N ranks itself (`python -m
torch.distributed.run --nproc-per-node N`, rendezvous on 127.0.0.1) BEFORE anything touches the GPU and relays rank 0's
JSON line.  Under a launcher (WORLD_SIZE set) it is one of the ranks; WORLD_SIZE != --gpus is an error, never a silent
1-GPU run.

A "step" is one full optimisation step of the hot path on one synthetic batch per rank: item gather, 8 HSTU layers,
4 prior heads, per-category sampled softmax over 8192 negatives, backward, RCCL gradient exchange, fused AdamW over all
parameters including the 454k x 256 item table.  Workload = BASELINE.json configs[1] (cfg1: L=200, P=8, D=256,
8 layers x 8 heads, C=4 prior heads, N=453 938 items, B=128 per GPU, bf16-mixed).  Inputs are generated on the device
before the timed region.  The timed region is pure wall clock (no events inside it); per-kernel HIP events are collected
in a SEPARATE pass afterwards.  Rank 0 prints ONE JSON line.

Extra objects on the line:
  roofline     - the dominant kernel of the train step (by summed HIP-event time of the evented pass), priced by its
                 ALGORITHMIC bytes against 8 TB/s when its arithmetic intensity is below the machine balance
                 (2.5 PFLOP/s / 8 TB/s = 312 flop/B), else by its ALGORITHMIC flops against the dense bf16 MFMA peak;
  gather       - the embedding gather's HBM roofline (north-star target >= 50 %);
  eval         - the eval leg (encoder -> heads -> full-catalog multi-head decode -> collector): users/s and the
                 catalog scorer's MFMA roofline (north-star target >= 40 %);
  cpu_baseline - the CPU oracle's train step (and eval step) on the host cores, one thread (the reference's own
                 setting, run.py:20-21) and all cores, on a bounded sample of the same workload.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "multi-head-recommendation-with-human-priors_amd", "code"))

MFMA_PEAK_TFLOPS = 2500.0     # dense bf16, MI355X_MICROARCH.md
HBM_PEAK_GBS = 8000.0
BALANCE = MFMA_PEAK_TFLOPS * 1e12 / (HBM_PEAK_GBS * 1e9)       # flop per byte at which the two roofs meet (312)


def parse(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--config", default="cfg1")
    ap.add_argument("--mode", default="train", choices=["train", "eval"])
    ap.add_argument("--batch", type=int, default=0, help="per-GPU batch override (0 = config)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-eval-leg", action="store_true", help="train mode: skip the eval sub-object")
    ap.add_argument("--no-host-probe", action="store_true", help="skip the 5 untimed steps that measure the host's issue time")
    ap.add_argument("--no-kernel-events", action="store_true", help="skip the evented pass (per-kernel HIP events)")
    ap.add_argument("--no-graph", action="store_true", help="issue every launch from the host (no hipGraph replay of the step)")
    ap.add_argument("--no-pack", action="store_true",
                    help="train batches without the loader's row-capacity hint: the encoder runs over all B x L window rows")
    ap.add_argument("--event-steps", type=int, default=10, help="steps of the evented pass (outside the timed region)")
    ap.add_argument("--pretrain-steps", type=int, default=0, help="eval mode: take this many train steps first (diagnostic)")
    ap.add_argument("--emulate-world", type=int, default=0,
                    help="one GPU: time the table-side kernels of the optimizer step (id sort, row buffer, segment sum + fix-up, lazy row "
                         "AdamW) on W ranks' worth of exchanged ids / rows - the part of a data-parallel step that does not shrink with W")
    return ap.parse_args(argv)


# ------------------------------------------------------------------------------------------------
# N > 1: start the ranks (before anything touches the GPU)
# ------------------------------------------------------------------------------------------------
def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def launch_ranks(args):
    """Start `args.gpus` ranks of this script under torch.distributed.run and return the launcher's exit code.  The ranks
    inherit stdout, so rank 0's JSON line is this process's output.  Nothing here imports torch.cuda or touches HIP."""
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "4")
    return subprocess.run(cmd, env=env).returncode


# ------------------------------------------------------------------------------------------------
# CPU baseline (oracle port; bounded sample)
# ------------------------------------------------------------------------------------------------
def _host_cores():
    try:
        cores = len(os.sched_getaffinity(0))
    except AttributeError:
        cores = os.cpu_count() or 1
    # the GPU box grants a CPU share (16 cores for one GPU) although os.cpu_count() reports the whole host:
    # oversubscribing 256 threads made one oracle step take minutes
    return max(1, min(cores, 16))


def _oracle_trainer(cfgd, item_num, B, pool_size=None):
    """-> train_step() -> seconds: one oracle train step (forward, autograd backward, dense AdamW over every parameter, fp32) of
    B windows at the shape of `cfgd`; pool_size: negatives per pool (default: the config's per-window count x B)."""
    import torch
    from oracle import hstu_oracle as HO
    from oracle import optim_oracle as OO
    import mhr_amd.synth as synth
    from REC.config.configurator import Config
    from REC.utils import get_model
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
    if pool_size is not None:
        data.config = dict(cfgd, num_negatives=-(-pool_size // B) * B)
    state = {k: (torch.zeros_like(p), torch.zeros_like(p)) for k, p in params.items()}
    step_no = [0]

    def train_step():
        batch = data.train_batch(B)
        t0 = time.perf_counter()
        out = HO.train_forward(w, ocfg, batch)
        out["loss"].backward()
        step_no[0] += 1
        with torch.no_grad():
            for k, p in params.items():
                if p.grad is None:
                    continue
                OO.adamw_step(p, p.grad, state[k][0], state[k][1], step_no[0], 1e-4)
                p.grad = None
        return time.perf_counter() - t0

    return train_step


def _median_rate(fn, threads, warm, timed, units):
    """warm untimed + timed calls of fn() -> seconds on `threads` threads; rate = units / MEDIAN step time."""
    import statistics
    import torch
    torch.set_num_threads(threads)
    for _ in range(warm):
        fn()
    ts = [fn() for _ in range(timed)]
    return round(units / max(statistics.median(ts), 1e-9), 3), len(ts)


def cpu_baseline_cfg0(warm=3, timed=10):
    """BASELINE.json configs[0] - the reference's own CPU-runnable case (HSTU-Pixel8M-base: seqlen 50, dim 64, 1 head) - as
    the oracle's train step on ONE thread (the reference pins one, run.py:20-21) and on all cores: `warm` warm-up steps, then
    the MEDIAN of `timed` steps (>= 3 + >= 10)."""
    import torch
    import mhr_amd.synth as synth
    spec = synth.CONFIGS["cfg0"]
    B = int(spec["cfg"].get("train_batch_size", 32))
    step = _oracle_trainer(dict(spec["cfg"]), spec["item_num"], B)
    cores = _host_cores()
    out = {}
    for name, thr in (("all_cores", cores), ("one_thread", 1)):
        v, n = _median_rate(step, thr, warm, timed, B)
        out[name] = {"value": v, "unit": "seq/s", "cores": thr, "steps": n, "warmup": warm, "statistic": "median"}
    torch.set_num_threads(cores)
    out["workload"] = f"cfg0 (reference's CPU case): L={spec['cfg']['MAX_ITEM_LIST_LENGTH']} D={spec['cfg']['item_embedding_size']} " \
                      f"{spec['cfg']['n_layers']} layers, N={spec['item_num']} items, B={B} windows per step"
    return out


def _reference_speed_ratio():
    """Container-side measurement (tools/ref_vs_oracle_cpu.py, committed under profiles/): how fast the reference's own
    hstu.py + collector.py run next to this oracle on identical weights and batches.  The reference cannot travel to the GPU
    box; this ties the port's numbers to it."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_ref_vs_oracle_cpu.json")))
    if not files:
        return None
    d = json.load(open(files[-1]))
    out = {"source": os.path.relpath(files[-1], ROOT), "host_cores_of_that_run": d.get("host_cores")}
    for case in ("cfg0", "cfg1_reduced"):
        for leg in ("train", "eval"):
            for thr, rec in (d.get(case, {}).get(leg, {}) or {}).items():
                if isinstance(rec, dict) and "oracle_over_reference_speed" in rec:
                    out[f"{case}_{leg}_{thr}_oracle_over_reference_speed"] = rec["oracle_over_reference_speed"]
    return out


def cpu_baseline(cfgd, item_num):
    """The oracle's fp32 train step (forward + autograd backward + dense AdamW over every parameter) and eval step
    (user heads -> full-catalog scores -> masks -> per-head top-k -> merge) on the host, at the GPU run's shape: same
    L, P, D, layers, heads, catalog and the SAME negatives per pool (8192); the sample is B = 4 windows per train step
    (the GPU batch is 128) and B = 16 users per eval step.  All cores first, then one thread (the reference pins one
    thread, run.py:20-21)."""
    import numpy as np
    import torch
    from oracle import decode_oracle as DO
    from oracle import hstu_oracle as HO
    from oracle import optim_oracle as OO
    import mhr_amd.synth as synth
    from REC.config.configurator import Config
    from REC.utils import get_model
    cores = _host_cores()
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
    n_pool = cfgd["num_negatives"]
    per_sample = -(-n_pool // B)                       # negatives per window so that the POOL has the GPU run's size
    data.config = dict(cfgd, num_negatives=per_sample * B)
    state = {k: (torch.zeros_like(p), torch.zeros_like(p)) for k, p in params.items()}
    step_no = [0]

    def train_step():
        batch = data.train_batch(B)
        t0 = time.perf_counter()
        out = HO.train_forward(w, ocfg, batch)
        out["loss"].backward()
        step_no[0] += 1
        with torch.no_grad():
            for k, p in params.items():
                if p.grad is None:
                    continue
                OO.adamw_step(p, p.grad, state[k][0], state[k][1], step_no[0], 1e-4)
                p.grad = None
        return time.perf_counter() - t0

    Be = 16
    K = max(cfgd["topk"])
    with torch.no_grad():
        feat = HO.compute_item_all({k: v.detach() for k, v in w.items()})
    tagm = data.item_tags.t().contiguous().numpy()                          # [C, N] bool
    S, C = cfgd["num_segment_head"], cfgd["num_prior_head"]

    def eval_step():
        eb = data.eval_batch(Be)
        t0 = time.perf_counter()
        with torch.no_grad():
            wd = {k: v.detach() for k, v in w.items()}
            u = HO.user_head_embeddings(wd, ocfg, eb[1])                    # [B, H, D] normalised
            sc = (u @ feat.T).numpy()                                       # [B, H, N] (hstu.py:965-979)
        if cfgd["loss"] == "prior":
            for h in range(sc.shape[1]):
                sc[:, h, ~tagm[h % C]] = -np.inf
        DO.suppress(sc, eb[3][0].numpy(), eb[3][1].numpy())
        v_, i_ = DO.per_head_topk(sc, K)                                    # collector.py:245
        DO.merge_dedup(v_, i_, K)                                           # collector.py:249-275
        return time.perf_counter() - t0

    # the metric's shape (cfg1) is a bounded sample - one step costs ~1.6 s on 16 threads, ~3.5 s on one: 2 warm-up + 5 timed
    # (median) on all cores, 1 + 2 on one thread; the eval step 2 + 6 / 1 + 3.  The reference's own CPU case (cfg0) gets the
    # full 3 + 10 protocol of BASELINE.md section 3.2 on both thread counts.
    tr_all, n1 = _median_rate(train_step, cores, 2, 5, B)
    tr_one, n2 = _median_rate(train_step, 1, 1, 2, B)
    ev_all, n3 = _median_rate(eval_step, cores, 2, 6, Be)
    ev_one, n4 = _median_rate(eval_step, 1, 1, 3, Be)
    torch.set_num_threads(cores)
    try:
        cfg0 = cpu_baseline_cfg0()
    except Exception as e:  # noqa: BLE001
        cfg0 = {"error": repr(e)[:160]}
    out = {"value": tr_all, "unit": "seq/s", "cores": cores, "kind": "port", "statistic": "median step time", "cfg0": cfg0,
           "one_thread": {"value": tr_one, "unit": "seq/s", "cores": 1, "steps": n2, "warmup": 1},
           "eval": {"value": ev_all, "unit": "users/s", "cores": cores, "steps": n3, "warmup": 2,
                    "one_thread": {"value": ev_one, "unit": "users/s", "cores": 1, "steps": n4, "warmup": 1}},
           "sample": f"2 warm-up + {n1} timed oracle train steps (fwd + autograd bwd + dense AdamW over all parameters, fp32) of {B} "
                     f"windows at the cfg shape with {per_sample * B} negatives per pool (= the GPU run's pool size; GPU batch is "
                     f"{cfgd['train_batch_size']} windows); eval: 2 + {n3} oracle eval steps of {Be} users over the full "
                     f"{item_num}-item catalog (scores, masks, per-head top-{K}, merge); cfg0: 3 + 10 steps of 64 windows"}
    ratio = _reference_speed_ratio()
    if ratio:
        out["reference_vs_port"] = ratio
    return out


# ------------------------------------------------------------------------------------------------
# roofline objects
# ------------------------------------------------------------------------------------------------
KERNEL_NAMES = {"mhr_nce_bwd_tokens": ["nce_bwd_rows_kernel"], "mhr_nce_bwd_negs": ["nce_bwd_n_kernel"],
                "mhr_hstu_attn_bwd": ["hstu_attn_bwd_kernel"], "mhr_hstu_attn_fwd": ["hstu_attn_fwd_kernel"],
                "mhr_nce_fwd": ["nce_fwd_d_kernel"],       # (the plain form launches no bit-table kernel; mhr_nce_fix_bits is its own call)
                "mhr_catalog_score_emit": ["catalog_emit_kernel"], "mhr_catalog_score_emit_sliced": ["catalog_emit_sliced_kernel"],
                "mhr_adam_rows": ["adam_rows_kernel"], "mhr_embedding_gather_fwd": ["gather_rows_kernel"], "mhr_embedding_gather_step": ["gather_step_kernel"]}


def _roof(kernel, flops, nbytes, ms, extra=None):
    """Roofline object of one launch: the binding roof is HBM when the arithmetic intensity (algorithmic flop per
    algorithmic byte) is below the machine balance, MFMA otherwise."""
    hbm = flops is None or (nbytes is not None and nbytes > 0 and flops / nbytes < BALANCE)
    if hbm:
        ach = nbytes / (ms * 1e-3) / 1e9
        r = {"bound": "hbm", "kernel": kernel, "achieved": round(ach, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
             "frac": round(ach / HBM_PEAK_GBS, 4), "traffic": None, "launch_ms": round(ms, 4),
             "algorithmic_bytes_per_launch": nbytes}
        if flops is not None:
            r["algorithmic_flops_per_launch"] = flops
            r["intensity_flop_per_byte"] = round(flops / nbytes, 1)
            r["mfma_TFLOPs"] = round(flops / (ms * 1e-3) / 1e12, 2)
    else:
        ach = flops / (ms * 1e-3) / 1e12
        r = {"bound": "mfma", "kernel": kernel, "achieved": round(ach, 2), "peak": MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
             "frac": round(ach / MFMA_PEAK_TFLOPS, 4), "traffic": None, "launch_ms": round(ms, 4),
             "algorithmic_flops_per_launch": flops}
        if nbytes:
            r["algorithmic_bytes_per_launch"] = nbytes
            r["intensity_flop_per_byte"] = round(flops / nbytes, 1)
    r.update(extra or {})
    return r


def _attach_traffic(roof, mode, standard_shape):
    """HBM bytes per launch of that kernel from the PMC passes (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE run on this same
    command and condensed by tools/summarize_profiles.py; counters cannot be read from inside the process)."""
    import glob
    knames = KERNEL_NAMES.get(roof["kernel"])
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", f"r*_hbm_traffic{'_eval' if mode == 'eval' else ''}.json")))
    if not (knames and files and standard_shape):
        return
    recs = [json.load(open(files[-1]))["kernels"].get(k) for k in knames]
    if recs[0]:
        key_ = "hbm_bytes_max_launch" if (mode == "eval" and "hbm_bytes_max_launch" in recs[0]) else "hbm_bytes_per_launch"
        roof["traffic"] = sum(r[key_] for r in recs if r)
        roof["traffic_source"] = os.path.relpath(files[-1], ROOT)


def _attach_rocprof(roof, kernel_name, flops=None, nbytes=None):
    """The same launch as rocprofv3 saw it (profiles/<round>_full_launch.json, condensed from the kernel trace of this command by
    tools/summarize_profiles.py): `rocprof_launch_ms` and the fraction that follows from it, next to the HIP-event figure of this
    run.  The two are different runs on possibly different boxes (the MFMA clock under load differs box to box)."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_full_launch.json")))
    if not files:
        return
    rec = json.load(open(files[-1]))["kernels"].get(kernel_name)
    if not rec:
        return
    ms = rec["median_ns"] * 1e-6
    roof["rocprof_launch_ms"] = round(ms, 4)
    if roof.get("bound") == "mfma" and flops:
        roof["frac_rocprof"] = round(flops / (ms * 1e-3) / 1e12 / MFMA_PEAK_TFLOPS, 4)
    elif nbytes:
        roof["frac_rocprof"] = round(nbytes / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)
    roof["rocprof_source"] = os.path.relpath(files[-1], ROOT)


def _gpu_clocks():
    """Current shader / memory clocks right after the timed legs, read from the card's sysfs files (`pp_dpm_sclk` / `pp_dpm_mclk`,
    the level marked `*`) - no child process: a process that has initialised the GPU must not exec another program on this pool
    (rocm-smi is a script: fork + exec).  Best effort: None where the files are not readable."""
    try:
        import torch
        pr = torch.cuda.get_device_properties(torch.cuda.current_device())
        dev = f"/sys/bus/pci/devices/{pr.pci_domain_id:04x}:{pr.pci_bus_id:02x}:{pr.pci_device_id:02x}.0"
        out = {}
        for name in ("sclk", "mclk"):
            with open(f"{dev}/pp_dpm_{name}") as f:
                cur = [ln.split(":", 1)[1].replace("*", "").strip() for ln in f if ln.rstrip().endswith("*")]
            out[name] = cur[0] if cur else None
        return out
    except Exception:  # noqa: BLE001
        return None


# ------------------------------------------------------------------------------------------------
def main():
    args = parse()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(launch_ranks(args))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but the launcher started WORLD_SIZE={world} ranks")
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    import torch
    import torch.distributed as dist
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the product path has no CPU fallback")
    # one process per GPU (the driver's launch).  MHR_DIST_BACKEND=gloo with several ranks sharing one card is only for
    # rehearsing the N > 1 code path on a single-GPU box.
    n_dev = torch.cuda.device_count()
    backend = os.environ.get("MHR_DIST_BACKEND", "nccl")
    if backend == "nccl" and world > n_dev:
        raise SystemExit(f"bench.py: {world} RCCL ranks need {world} GPUs, this node shows {n_dev}")
    torch.cuda.set_device(local_rank % n_dev)
    dev = torch.device("cuda", local_rank % n_dev)
    if world > 1:
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)
    elif os.environ.get("MHR_FORCE_DP", "0") == "1":
        # one-GPU rehearsal of the data-parallel step on RCCL itself: a process group of ONE rank, every collective of the exchange
        # a real RCCL call issued between the replayed graph segments (mhr_amd.distributed.FORCE_DP)
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", str(_free_port()))
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    import mhr_amd.synth as synth
    from mhr_amd import ops
    from REC.config.configurator import Config, apply_run_fixups
    from REC.trainer import Trainer
    from REC.utils import get_model

    spec = synth.CONFIGS[args.config]
    cfgd = dict(spec["cfg"], device=dev, total_iters=30000, eval_interval=0, checkpoint_dir=None, save_model_note="bench")
    if args.batch:
        cfgd["train_batch_size"] = cfgd["eval_batch_size"] = args.batch
    if args.no_graph:
        cfgd["hip_graph"] = False
    elif args.mode == "train" and world == 1:
        cfgd["hip_graph_required"] = True      # a capture that fails is an error here, not a silent host-issued run
        # (N > 1: the replayed data-parallel step has only ever run as a two-rank gloo rehearsal on one card - on real RCCL
        #  hardware a failed capture degrades to host-issued launches; the line says so in graph_active / graph_failure)
    cfg = apply_run_fixups(Config(config_dict=cfgd))
    N = spec["item_num"]
    data = synth.SyntheticData(cfg, N, dev, seed=2020, rank=rank, world=world)
    # the loader's host-side count of a batch's valid positions, as a bucketed static capacity on the mask (synth.rows_capacity):
    # with it the encoder runs over the valid rows only (packed rows, csrc/rows_pack.hip); batches are generated outside the
    # timed region, as a CPU collate would produce the count for free
    # (data parallel: every rank must see the same capacity for a step - run_leg takes the maximum over the ranks)
    data.rows_hint = not args.no_pack
    cfg["int_to_category"] = data.int_to_category
    torch.manual_seed(2020)
    model = get_model("HSTU")(cfg, data).to(dev)
    trainer = Trainer(cfg)
    trainer.setup_model(model)
    trainer.train_step = 3000            # past the warm-up so lr > 0: every step really moves the parameters
    if data.rows_hint and not (model._packable(cfg["MAX_ITEM_LIST_LENGTH"]) and model._dead_rows_allowed()):
        data.rows_hint = False           # (a model that cannot run on packed rows: no hint, one step graph)
    L, P, D, C = cfg["MAX_ITEM_LIST_LENGTH"], cfg["pred_len"], cfg["hstu_embedding_size"], data.C
    H = model.medusa_num_heads
    standard = not args.batch

    def sync():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def max_over_ranks(x):
        if world > 1:
            t = torch.tensor([x], dtype=torch.float64, device=dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            return float(t)
        return x

    EVENTED = ["mhr_nce_fwd", "mhr_nce_bwd_tokens", "mhr_nce_bwd_negs", "mhr_nce_shared_fwd_tokens", "mhr_nce_shared_bwd_rows",
               "mhr_nce_shared_bwd_targets", "mhr_nce_shared_bwd_tokens", "mhr_catalog_score_emit", "mhr_catalog_score_emit_sliced",
               "mhr_hstu_attn_fwd", "mhr_hstu_attn_bwd", "mhr_adam_rows", "mhr_adam_rows_lazy", "mhr_embedding_gather_fwd",
               "mhr_sparse_rows_segment_sum", "mhr_topk_select", "mhr_topk_select_sliced", "mhr_sum_rows_into", "mhr_embedding_gather_step"]

    def run_leg(mode, warmup, steps, event_steps):
        """-> dict(elapsed, host_enqueue, host_alone, prof, prof_max, last, batches, B).  Wall-clock pass first (nothing but
        the steps between the two syncs), then - outside it - the evented pass and the idle-queue host probe."""
        B = cfg["train_batch_size"] if mode == "train" else cfg["eval_batch_size"]
        n_steps = warmup + steps
        if mode == "train":
            batches = [data.train_batch(B) for _ in range(min(n_steps, 8))]
            if world > 1 and getattr(data, "rows_hint", False):
                # one capacity per step for ALL ranks (the largest): the step graphs are keyed on it and the ranks must capture /
                # replay in lockstep - a capturing rank issues no collective
                caps = torch.tensor([int(bt[2]._mhr_rows_cap) for bt in batches], dtype=torch.int64, device=dev)
                dist.all_reduce(caps, op=dist.ReduceOp.MAX)
                for bt, c_ in zip(batches, caps.tolist()):
                    bt[2]._mhr_rows_cap = int(c_)
            model.train()

            def step(i, eager=False):
                return trainer.train_step_fn(batches[i % len(batches)], **({"graph": False} if eager else {}))
        else:
            batches = [data.eval_batch(B) for _ in range(min(n_steps, 4))]
            trainer.compute_item_feature(data.item_tags)

            def step(i, eager=False):
                fused, pu, pi, tt, _ = trainer._full_sort_batch_eval(batches[i % len(batches)], graph=False if eager else None)
                trainer.eval_collector.eval_batch_collect(fused, pu, pi)
                return None
        ops.PROFILE = None
        if mode == "train" and not args.no_graph:
            # one captured step graph per batch signature (shapes + the packed encoder's row capacity): every signature gets its
            # three host-issued steps and its capture here, in front of the W warm-up steps - the timed region then only replays
            seen_caps = {}
            for bi, bt in enumerate(batches):
                seen_caps.setdefault(getattr(bt[2], "_mhr_rows_cap", None), bi)
            for bi in seen_caps.values():
                for _ in range(4):
                    step(bi)
        for i in range(warmup):
            step(i)
        sync()
        t0 = time.perf_counter()
        last = None
        for i in range(steps):
            last = step(warmup + i)
        if mode == "train" and getattr(trainer.optimizer, "lazy", False):
            # lazy table optimizer: rows without gradient are replayed when next read.  Whatever is still deferred at the end of
            # the timed region is paid INSIDE it (one flush of the whole table): no optimizer work of these K steps is left out.
            trainer.optimizer.flush_table()
        host_enqueue = time.perf_counter() - t0          # the host is done issuing; the GPU may still be working
        sync()
        elapsed = max_over_ranks(time.perf_counter() - t0)
        host_alone = []                                   # host cost of issuing ONE step into an idle queue
        for i in range(0 if args.no_host_probe else 5):
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            step(n_steps + i)
            host_alone.append(time.perf_counter() - t1)
        sync()
        host_alone = sorted(host_alone)[len(host_alone) // 2] if host_alone else float('nan')
        prof, prof_max = {}, {}
        if event_steps > 0 and not args.no_kernel_events:
            # evented pass: every C-ABI call of EVENTED bracketed by HIP events on its launch stream.  Host-issued launches
            # (a replayed hipGraph has no per-kernel events), a few steps, outside the timed region.
            ops.PROFILE = {k: [] for k in EVENTED}
            for i in range(event_steps):
                step(n_steps + 5 + i, eager=True)
            sync()
            for k_ in list(ops.PROFILE):   # median over the steps of each step's longest launch (robust against one outlier)
                raw = sorted(ops.profile_raw(k_), reverse=True)[:event_steps]
                if raw:
                    prof_max[k_] = raw[len(raw) // 2]
            prof = ops.profile_summary()
            ops.PROFILE = None
        return dict(elapsed=elapsed, host_enqueue=host_enqueue, host_alone=host_alone, prof=prof, prof_max=prof_max,
                    last=last, batches=batches, B=B, steps=steps, event_steps=max(1, event_steps))

    def catalog_roofline(leg, Bev):
        """The launch that scores the whole catalog (the two threshold-sample launches of a step score 1/14 + 1/222 of it and
        are latency-sized): algorithmic flops of that launch / its own duration (the longest of the step)."""
        for name in ("mhr_catalog_score_emit_sliced", "mhr_catalog_score_emit"):
            if name in leg["prof"] and leg["prof"][name][0] > 0:
                launches, mean_ms, _ = leg["prof"][name]
                flops = 2.0 * Bev * H * D * N
                nbytes = N * D * 2.0 + Bev * H * D * 2.0          # the bf16 catalog once + the user rows
                full_ms = leg["prof_max"].get(name, mean_ms)
                r = _roof(name, flops, nbytes, full_ms,
                          {"launches_per_step": launches / leg["event_steps"],
                           "all_launches_ms_per_step": round(mean_ms * launches / leg["event_steps"], 4)})
                _attach_traffic(r, "eval", standard and args.config == "cfg1")
                if standard and args.config == "cfg1" and Bev == 256:
                    _attach_rocprof(r, "catalog_emit_sliced_kernel", flops=flops)
                return r
        return None

    def train_roofline(leg):
        """Roofline of the dominant kernel of the train step (largest summed HIP-event time with a roofline model)."""
        prof, B = leg["prof"], leg["B"]
        es = leg["event_steps"]
        batches = leg["batches"]
        order = sorted(prof.items(), key=lambda kv: -kv[1][2])
        # The sampled softmax's two streaming kernels take the same time to within a few per cent; which of them is "largest" flips
        # from box to box.  While they are within 5 % of each other the line prices the FORWARD (as every earlier round's line did)
        # and carries the negative-side backward beside it (`roofline_pair`), so that the fraction stays comparable run to run.
        pair = [kv for kv in order if kv[0] in ("mhr_nce_fwd", "mhr_nce_bwd_negs")]
        if order and len(pair) == 2 and order[0][0] == "mhr_nce_bwd_negs" and pair[1][1][2] >= 0.95 * pair[0][1][2]:
            order = [pair[1], pair[0]] + [kv for kv in order if kv[0] not in ("mhr_nce_fwd", "mhr_nce_bwd_negs")]
        second = None
        for name, (launches, mean_ms, total_ms) in order:
            if name in ("mhr_nce_fwd", "mhr_nce_bwd_negs"):
                # ONE launch serves all groups (prior categories).  The P prediction offsets of a position share one query
                # row (reference hstu.py:682-690: head c's embedding at (b, l) is the query of tokens (b, l, 0..P-1)), so the
                # products against the negative pool are per DISTINCT ROW: algorithmic flop per launch = forward 2 products
                # (logits s, token-side gradient numerator U) x 2 rows N_neg D, negative-side backward 1 product (dN) - its
                # recomputation of s is an implementation cost, not algorithmic work.  The reference's own per-token
                # evaluation (3 x 2 N_tok N_neg D forward) is reported beside it; most of it is repeats.
                items, _, mask, tags = batches[-1]
                idx = torch.arange(L, device=dev)[None, :] + 1 + torch.arange(P, device=dev)[:, None]
                mb = mask.bool()
                valid = mb[:, None, :L] & mb[:, idx]
                if cfg["loss"] == "prior":
                    live = valid[..., None] & tags[:, idx].bool()                          # [B, P, L, C]
                else:
                    live = valid[..., None]
                n_tok, n_row = float(live.sum()), float(live.any(dim=1).sum())
                shared = os.environ.get("MHR_NCE_SHARE_ROWS", "1") != "0" and P > 1
                units = n_row if shared else n_tok
                n_neg = world * B * data.n_neg(B)
                fwd = name == "mhr_nce_fwd"
                flops = (4.0 if fwd else 2.0) * units * n_neg * D
                r = _roof(name, flops, None, mean_ms,
                          {"launches_per_step": launches / es, "tokens_per_launch": n_tok, "distinct_query_rows_per_launch": n_row,
                           "row_sharing": shared, "negatives_per_group": n_neg,
                           "executed_flops_per_launch": 4.0 * units * n_neg * D,
                           "reference_per_token_formulation_flops": (6.0 if fwd else 2.0) * n_tok * n_neg * D})
            elif name in ("mhr_hstu_attn_fwd", "mhr_hstu_attn_bwd"):
                # one launch = one layer over the B sequences.  Algorithmic work: forward 4 L^2 D flop per
                # (sequence, layer), backward twice that; algorithmic bytes 4 L D 2 B forward (q, k, v in, out),
                # 7 L D 2 B backward (q, k, v, dO in; dq, dk, dv out).  Intensity 8L/14 = 114 flop/B at L = 200: HBM roof.
                bwd = name.endswith("bwd")
                flops = (8.0 if bwd else 4.0) * B * L * L * D
                nbytes = (7.0 if bwd else 4.0) * B * L * D * 2.0
                r = _roof(name, flops, nbytes, mean_ms, {"launches_per_step": launches / es})
            elif name == "mhr_adam_rows":
                r = _roof(name, None, N * D * 24.0, mean_ms, {"launches_per_step": launches / es})
            else:
                continue
            _attach_traffic(r, "train", standard and args.config == "cfg1")
            if standard and args.config == "cfg1" and name in KERNEL_NAMES:
                _attach_rocprof(r, KERNEL_NAMES[name][0], flops=r.get("algorithmic_flops_per_launch"), nbytes=r.get("algorithmic_bytes_per_launch"))
            if second is None and name in ("mhr_nce_fwd", "mhr_nce_bwd_negs") and len(pair) == 2:
                second = r                       # first of the pair: keep going for its partner
                continue
            if second is not None:
                if name in ("mhr_nce_fwd", "mhr_nce_bwd_negs"):
                    second["roofline_pair"] = {k: r[k] for k in ("kernel", "achieved", "frac", "launch_ms", "algorithmic_flops_per_launch",
                                                                 "executed_flops_per_launch", "frac_rocprof") if k in r}
                    return second
                continue
            return r
        return second

    def emulate_world(Wem, iters=10):
        """What every rank does with the exchanged gradient rows of W ranks (reference: the dense 116 M-parameter all-reduce +
        FusedAdam of trainer.py:292-299, 532-536; here SURVEY 8e's sparse exchange): sort W x B (L + P) private ids + the shared
        negative ids, zero the row buffer, segment-sum (bf16 private rows off the wire, fp32 shared block), lazy AdamW on the
        touched rows.  Unmeasured on multi-GPU hardware: this is the same kernels on one card fed the W-rank id / row volume."""
        opt = trainer.optimizer
        B = cfg["train_batch_size"]
        n_pools = C if (cfg["loss"] == "prior" and cfg.get("neg_sample_by_cat")) else 1
        n_priv, n_sh = B * (L + P), n_pools * cfg["num_negatives"]
        if model._row_slot is None:
            model._row_slot = torch.full((N,), -1, dtype=torch.int32, device=dev)
        res = {}
        for w_ in sorted({1, Wem}):
            ids = torch.cat([data._zipf((w_ * n_priv,)), data._uniform_items((n_sh,))]).contiguous()
            rows_priv = (torch.randn(w_ * n_priv, D, device=dev) * 1e-3).to(torch.bfloat16 if w_ > 1 else torch.float32)
            rows_sh = torch.randn(n_sh, D, device=dev) * 1e-3
            ev = [[torch.cuda.Event(enable_timing=True) for _ in range(5)] for _ in range(iters)]
            for it in range(iters + 2):
                e = ev[max(0, it - 2)]
                e[0].record()
                sorted_ids, perm = torch.sort(ids)
                e[1].record()
                out_rows = torch.zeros(ids.numel(), D, dtype=torch.float32, device=dev)
                e[2].record()
                ops.sparse_rows_segment_sum(sorted_ids, perm, rows_priv, rows_sh, None, 0, 0, out_rows, model._row_slot)
                e[3].record()
                opt.step_count += 1
                opt._push_consts(opt.step_count, 1e-4)
                opt._lazy_call(1, sorted_ids, out_rows, model._row_slot, 1.0 / w_)
                if opt.step_count % 64 == 0:
                    opt._lagging = True
                    opt.flush_table()
                e[4].record()
            torch.cuda.synchronize()
            med = lambda a, b: sorted(x[a].elapsed_time(x[b]) for x in ev)[iters // 2]   # noqa: E731
            uniq = int(torch.unique(ids).numel())
            res[f"W{w_}"] = {"ids": int(ids.numel()), "unique_ids": uniq, "sort_ms": round(med(0, 1), 4), "zero_rows_ms": round(med(1, 2), 4),
                            "segment_sum_ms": round(med(2, 3), 4), "adam_rows_lazy_ms": round(med(3, 4), 4), "total_ms": round(med(0, 4), 4)}
        opt._lagging = True
        opt.flush_table()
        res["note"] = ("one card, synthetic W-rank volume (Zipf(1.05) private ids per rank, bf16 private rows as on the wire, fp32 shared "
                       "block); no collective runs - multi-GPU behaviour stays unmeasured until a SCALE record exists")
        return res

    rc_fail = None
    legs = {}
    if args.mode == "eval" and args.pretrain_steps:
        model.train()
        pb = [data.train_batch(cfg["train_batch_size"]) for _ in range(4)]
        for i in range(args.pretrain_steps):
            trainer.train_step_fn(pb[i % 4])
        torch.cuda.synchronize()
    main_leg = legs[args.mode] = run_leg(args.mode, args.warmup, args.steps, args.event_steps)
    graph_expected = bool(args.mode == "train" and not args.no_graph and (trainer._graph_ok(main_leg["batches"][0]) or getattr(trainer, "_graph_failed", False)))
    graph_on = bool(getattr(trainer, "graph_active", False))
    if args.mode == "train" and not args.no_eval_leg:
        legs["eval"] = run_leg("eval", 3, 10, min(args.event_steps, 5))

    if rank == 0:
        B = main_leg["B"]
        elapsed = main_leg["elapsed"]
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
                                      if getattr(trainer.optimizer, "lazy", False) and args.mode == "train" else "")
                                   + ("; step replayed from a hipGraph" if graph_on and args.mode == "train" else "")
                                   + ("; encoder on packed rows (the loader's row capacity, one step graph per capacity)"
                                      if getattr(data, "rows_hint", False) and args.mode == "train" else ""),
                       "global_batch": world * B, "seq_len": L, "parallelism": f"dp{world}"},
        }
        if getattr(data, "rows_hint", False) and args.mode == "train":
            caps = sorted({int(getattr(bt[2], "_mhr_rows_cap", 0) or 0) for bt in main_leg["batches"]})
            out["packed_rows"] = {"window_rows_per_batch": B * L, "capacities": caps,
                                  "mean_valid_rows": round(float(sum(float(bt[2][:, :L].sum()) for bt in main_leg["batches"])) / max(1, len(main_leg["batches"])), 1)}
        assert out["n_gpus"] == args.gpus
        out["graph_active"] = graph_on if args.mode == "train" else any(
            g_.graph is not None for g_ in trainer.__dict__.get("_eval_graphs", {}).values())
        out["graph_expected"] = bool(graph_expected)
        if getattr(trainer, "graph_failure", None):
            out["graph_failure"] = trainer.graph_failure
        out["host_enqueue_ms_per_step"] = round(1000 * main_leg["host_enqueue"] / args.steps, 3)
        st_ = getattr(trainer, "_step_graph", None)
        if (world > 1 or os.environ.get("MHR_FORCE_DP", "0") == "1") and st_ is not None and st_.graph is not None and st_.n:
            # data parallel: the replay plan's host calls ARE the collectives (issue + stream waits); with the gloo rehearsal backend
            # a wait blocks the host until the GPU has produced the data, with RCCL it only orders streams
            out["host_in_collectives_ms_per_step"] = round(1000 * st_.graph.host_s / max(1, st_.graph.n_replays), 3)
            out["graph_segments_per_step"] = sum(1 for k_, _ in st_.graph.plan if k_ in ("main", "side"))
            out["collective_host_calls_per_step"] = sum(1 for k_, _ in st_.graph.plan if k_ == "host")
        if main_leg["host_alone"] == main_leg["host_alone"]:
            out["host_issue_ms_idle_queue"] = round(1000 * main_leg["host_alone"], 3)
        last = main_leg["last"]
        if last is not None and args.mode == "train":
            out["loss"] = round(float(last["loss"].detach()), 4)
        prof = main_leg["prof"]
        es = main_leg["event_steps"]
        if prof:
            roof = train_roofline(main_leg) if args.mode == "train" else catalog_roofline(main_leg, B)
            if roof:
                out["roofline"] = roof
            out["kernel_ms_per_step"] = {k: round(v[2] / es, 3) for k, v in sorted(prof.items(), key=lambda kv: -kv[1][2])}
            if args.mode == "train" and "mhr_nce_fwd" in prof and cfg["loss"] == "prior" and "mhr_nce_bwd_negs" in prof:
                # sampled softmax: ALGORITHMIC = the reference's per-token formulation (3 products forward, dN
                # backward) - with query-row sharing most of those products are never executed, so that figure is NOT a
                # roofline fraction; EXECUTED = what the row-sharing kernels run (2 products per distinct query row each way)
                items_, _, mask_, tags_ = main_leg["batches"][-1]
                idx_ = torch.arange(L, device=dev)[None, :] + 1 + torch.arange(P, device=dev)[:, None]
                mb_ = mask_.bool()
                live_ = (mb_[:, None, :L] & mb_[:, idx_])[..., None] & tags_[:, idx_].bool()                 # [B, P, L, C]
                n_tok_, n_row_ = float(live_.sum()), float(live_.any(dim=1).sum())
                n_neg_ = world * B * data.n_neg(B)
                shared = os.environ.get("MHR_NCE_SHARE_ROWS", "1") != "0" and P > 1
                units = n_row_ if shared else n_tok_
                fe = 4.0 * units * n_neg_ * D / (prof["mhr_nce_fwd"][1] * 1e-3) / 1e12
                be = 4.0 * units * n_neg_ * D / (prof["mhr_nce_bwd_negs"][1] * 1e-3) / 1e12
                out["sampled_softmax"] = {
                    "tokens": n_tok_, "distinct_query_rows": n_row_, "row_sharing": shared,
                    "fwd_ms": round(prof["mhr_nce_fwd"][1], 4), "bwd_negs_ms": round(prof["mhr_nce_bwd_negs"][1], 4),
                    "fwd_executed_TFLOPs": round(fe, 1), "fwd_executed_frac_of_mfma_peak": round(fe / MFMA_PEAK_TFLOPS, 4),
                    "bwd_negs_executed_TFLOPs": round(be, 1), "bwd_negs_executed_frac_of_mfma_peak": round(be / MFMA_PEAK_TFLOPS, 4),
                    "reference_formulation_TFLOPs_equiv": round(6.0 * n_tok_ * n_neg_ * D / (prof["mhr_nce_fwd"][1] * 1e-3) / 1e12, 1),
                    "mfma_peak_TFLOPs": MFMA_PEAK_TFLOPS}
            if ("mhr_embedding_gather_fwd" in prof or "mhr_embedding_gather_step" in prof) and args.mode == "train":
                # item ids + negative-pool ids: table rows read and written as fp32, plus the position-added encoder input
                n_pools = C if (cfg["loss"] == "prior" and cfg.get("neg_sample_by_cat")) else 1
                n_item, n_negs = B * (L + P), data.n_neg(B) * B * n_pools * world
                fused = "mhr_embedding_gather_step" in prof        # one launch: item windows + negative pools normalised to bf16
                ms = prof["mhr_embedding_gather_step" if fused else "mhr_embedding_gather_fwd"][1]
                # bytes the launch moves: item windows read fp32, written fp32 (targets) + fp32 position-added encoder input;
                # negative pools read fp32, written normalised bf16 (+ 4 B norm per row) - or fp32 rows without the fusion
                own = (n_item * 2 + B * L) * D * 4.0 + ((n_negs * D * 6.0 + n_negs * 4.0) if fused else n_negs * D * 8.0)
                # SURVEY 8d / BASELINE.md contract: rows x D x (4 B table read + 2 B bf16 out) over ALL gathered rows
                contract = (n_item + n_negs) * D * 6.0
                g = _roof("mhr_embedding_gather_step" if fused else "mhr_embedding_gather_fwd", None, own, ms,
                          {"rows_per_launch": n_item + n_negs, "item_rows": n_item, "negative_rows": n_negs,
                           "contract_bytes_per_launch": contract, "contract_GBps": round(contract / (ms * 1e-3) / 1e9, 1),
                           "frac_on_contract_bytes": round(contract / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                           "note": "frac = the bytes this launch moves (fp32 residual stream: item rows leave as fp32 twice, negatives as "
                                   "normalised bf16) / time / 8 TB/s; frac_on_contract_bytes prices the same time on SURVEY 8d's rows x D x 6 B"})
                _attach_traffic(g, "train", standard and args.config == "cfg1")
                out["gather"] = g
            if "mhr_adam_rows" in prof:
                out["adam_rows_GBps"] = round(N * D * 24.0 / (prof["mhr_adam_rows"][1] * 1e-3) / 1e9, 1)
        if "eval" in legs and args.mode == "train":
            ev = legs["eval"]
            Bev = ev["B"]
            eo = {"value": round(world * Bev * ev["steps"] / ev["elapsed"], 1), "unit": "users/s", "steps": ev["steps"], "warmup": 3,
                  "ms_per_step": round(1000 * ev["elapsed"] / ev["steps"], 3), "batch_per_gpu": Bev,
                  "workload": f"encoder -> {H} heads -> full-catalog decode over N={N} items (tag / pad / history masks, exact per-head "
                              f"top-{max(cfg['topk'])}) -> merge / dedup / hit matrix"}
            cr = catalog_roofline(ev, Bev)
            if cr:
                eo["roofline"] = cr
            eo["kernel_ms_per_step"] = {k: round(v[2] / ev["event_steps"], 3) for k, v in sorted(ev["prof"].items(), key=lambda kv: -kv[1][2])}
            eo["host_enqueue_ms_per_step"] = round(1000 * ev["host_enqueue"] / ev["steps"], 3)
            eo["graph_active"] = any(g_.graph is not None for g_ in trainer.__dict__.get("_eval_graphs", {}).values())
            out["eval"] = eo
        if args.emulate_world > 1 and world == 1 and args.mode == "train" and getattr(trainer.optimizer, "lazy", False):
            out["emulated_world"] = emulate_world(args.emulate_world)
        clk = _gpu_clocks()
        if clk:
            out["gpu_clocks_after_run"] = clk
        if not args.no_cpu_baseline and world == 1:
            try:
                out["cpu_baseline"] = cpu_baseline(dict(spec["cfg"]), N)
            except Exception as e:  # noqa: BLE001 - the baseline must not kill the bench line
                out["cpu_baseline"] = {"value": None, "error": repr(e)[:200]}
        print(json.dumps(out), flush=True)
        if out.get("graph_expected") and not out["graph_active"] and world == 1:
            rc_fail = "bench.py: the train step was expected to replay from a hipGraph and did not"
    if dist.is_initialized():
        dist.barrier()
        dist.destroy_process_group()
    if rc_fail:
        raise SystemExit(rc_fail)


if __name__ == "__main__":
    main()
