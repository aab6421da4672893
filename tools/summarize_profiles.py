"""Condense rocprofv3 output (gpurun_out/…) into the small, tracked summaries under profiles/.

    python tools/summarize_profiles.py r01

reads   gpurun_out/prof_<round>_{train,eval}/*/*_kernel_stats.csv      (rocprofv3 --kernel-trace --stats)
        gpurun_out/pmc_<round>_{fetch,write}/*/*_counter_collection.csv (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE)
writes  profiles/<round>_{train,eval}_kernel_stats.csv   top kernels, names shortened
        profiles/<round>_hbm_traffic.json                 per-kernel HBM bytes per launch, corrected as
                                                          MI355X_MICROARCH.md prescribes (FETCH_SIZE is in KB and
                                                          counts 64 B per 128-B request on gfx950 -> x2; WRITE_SIZE KB x1)
"""
import csv
import glob
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
csv.field_size_limit(1 << 30)


def short(name):
    m = re.search(r"([a-z][a-z0-9_]*?_kernel)(?![a-z0-9_])|([a-z][a-z0-9_]*?_kernel)I", name)
    if m and "at::native" not in name and "rocprim" not in name:
        return m.group(1) or m.group(2)
    if name.startswith("Cijk_"):
        m = re.search(r"(Cijk_\w+?_MT\d+x\d+x\d+)", name)
        return (m.group(1) if m else name[:60]) + " (hipBLASLt)"
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    return name[:110]


def newest(pattern):
    files = sorted(glob.glob(pattern), key=os.path.getmtime)
    return files[-1] if files else None


def kernel_stats(rnd, leg, top=40):
    f = newest(os.path.join(ROOT, "gpurun_out", f"prof_{rnd}_{leg}", "*", "*_kernel_stats.csv"))
    if f is None:
        return
    rows = list(csv.DictReader(open(f)))
    out = os.path.join(ROOT, "profiles", f"{rnd}_{leg}_kernel_stats.csv")
    with open(out, "w", newline="") as fh:
        w = csv.writer(fh)
        w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs"])
        for r in rows[:top]:
            w.writerow([short(r["Name"]), r["Calls"], r["TotalDurationNs"], f'{float(r["AverageNs"]):.0f}', r["Percentage"],
                        r["MinNs"], r["MaxNs"]])
        rest = rows[top:]
        if rest:
            w.writerow([f"({len(rest)} more kernels)", sum(int(r["Calls"]) for r in rest),
                        sum(int(r["TotalDurationNs"]) for r in rest), "", f'{sum(float(r["Percentage"]) for r in rest):.2f}', "", ""])
    print("wrote", out)


def traffic(rnd, mode=""):
    acc = {}
    for leg, counter, scale in (("fetch", "FETCH_SIZE", 2.0 * 1024), ("write", "WRITE_SIZE", 1024.0)):
        f = newest(os.path.join(ROOT, "gpurun_out", f"pmc_{rnd}_{mode}{leg}", "*", "*_counter_collection.csv"))
        if f is None:
            continue
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] != counter:
                continue
            k = short(r["Kernel_Name"])
            if "at::native" in k or "rocprim" in k or "hipBLASLt" in k:
                continue
            e = acc.setdefault(k, {"fetch": [], "write": []})
            e[leg].append(float(r["Counter_Value"]) * scale)
    out = {}
    for k, e in sorted(acc.items()):
        fb = sum(e["fetch"]) / max(1, len(e["fetch"]))
        wb = sum(e["write"]) / max(1, len(e["write"]))
        out[k] = {"launches_sampled": len(e["fetch"]), "hbm_read_bytes_per_launch": round(fb), "hbm_write_bytes_per_launch": round(wb),
                  "hbm_bytes_per_launch": round(fb + wb),
                  # the largest launch of the kernel (eval: the full-catalog pass next to the two small sample passes)
                  "hbm_bytes_max_launch": round(max(e["fetch"], default=0) + max(e["write"], default=0))}
    if out:
        p = os.path.join(ROOT, "profiles", f"{rnd}_hbm_traffic{'_eval' if mode else ''}.json")
        json.dump({"how": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes of `python bench.py " + ("--mode eval " if mode else "") + "--no-cpu-baseline "
                          "--steps 4 --warmup 2`; FETCH_SIZE KB x 2 (gfx950 tallies 128-B requests at 64 B), WRITE_SIZE KB x 1; "
                          "mean over all launches of the kernel", "kernels": out}, open(p, "w"), indent=1)
        print("wrote", p)


def mfma_util(rnd, mode=""):
    """MFMA utilisation per kernel from `rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE` on bench.py
    (SURVEY 8d).  SQ_VALU_MFMA_BUSY_CYCLES counts matrix-pipe cycles summed over the chip's 1024 SIMDs (32 per
    v_mfma_f32_32x32x16_bf16, MI355X_MICROARCH.md); GRBM_GUI_ACTIVE is summed over the 8 XCDs, so GRBM / 8 is the launch's
    length in shader cycles:  util = MFMA_BUSY / (1024 x GRBM / 8)."""
    f = newest(os.path.join(ROOT, "gpurun_out", f"pmc_{rnd}_{mode}mfma", "*", "*_counter_collection.csv"))
    if f is None:
        return
    per = {}
    for r in csv.DictReader(open(f)):
        k = short(r["Kernel_Name"])
        per.setdefault((k, r["Dispatch_Id"]), {})[r["Counter_Name"]] = float(r["Counter_Value"])
    acc = {}
    for (k, _), c in per.items():
        if c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) <= 0 or c.get("GRBM_GUI_ACTIVE", 0) <= 0:
            continue
        acc.setdefault(k, []).append(c)
    out = {}
    for k, lst in acc.items():
        big = max(lst, key=lambda c: c["SQ_VALU_MFMA_BUSY_CYCLES"])         # the kernel's largest launch (eval: the full-catalog pass)
        tot_m = sum(c["SQ_VALU_MFMA_BUSY_CYCLES"] for c in lst)
        tot_g = sum(c["GRBM_GUI_ACTIVE"] for c in lst)
        out[k] = {"launches_sampled": len(lst), "mfma_busy_cycles_per_launch": round(tot_m / len(lst)),
                  "shader_cycles_per_launch": round(tot_g / 8 / len(lst)),
                  "mfma_util": round(tot_m / (1024.0 * tot_g / 8.0), 4),
                  "mfma_util_largest_launch": round(big["SQ_VALU_MFMA_BUSY_CYCLES"] / (1024.0 * big["GRBM_GUI_ACTIVE"] / 8.0), 4),
                  "sq_busy_cu_cycles_per_launch": round(sum(c.get("SQ_BUSY_CU_CYCLES", 0) for c in lst) / len(lst))}
    if out:
        p = os.path.join(ROOT, "profiles", f"{rnd}_mfma_util{'_eval' if mode else ''}.json")
        json.dump({"how": "rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE -- python bench.py " +
                          ("--mode eval " if mode else "") + "(its own pass, no tracing domains); util = MFMA_BUSY / (1024 SIMDs x GRBM_GUI_ACTIVE / 8 XCDs)",
                   "kernels": dict(sorted(out.items(), key=lambda kv: -kv[1]["mfma_busy_cycles_per_launch"]))}, open(p, "w"), indent=1)
        print("wrote", p)


def full_launches(rnd):
    """Per-launch durations of the kernels whose bench roofline is priced on ONE launch of a step, from the rocprofv3 kernel
    TRACE of the same command (the --stats table averages a kernel's launches: the catalog scorer runs three per eval step, two
    of them small threshold-sample passes).  For each kernel: the median of the step's LARGEST launch -> profiles/<rnd>_full_launch.json;
    bench.py prints it next to its own HIP-event figure (`rocprof_launch_ms`, `frac_rocprof`)."""
    out = {}
    for leg, names, per_step in (("eval", ("catalog_emit_sliced_kernel", "catalog_emit_wide_kernel"), 3),
                                 ("train", ("nce_fwd_d_kernel", "nce_bwd_n_kernel", "gather_step_kernel", "hstu_attn_bwd_kernel",
                                            "hstu_attn_fwd_kernel"), None)):
        f = newest(os.path.join(ROOT, "gpurun_out", f"prof_{rnd}_{leg}", "*", "*_kernel_trace.csv"))
        if f is None:
            continue
        durs = {}
        for r in csv.DictReader(open(f)):
            k = short(r["Kernel_Name"])
            if k in names:
                durs.setdefault(k, []).append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
        for k, d in durs.items():
            d.sort(reverse=True)
            top = d[:max(1, len(d) // per_step)] if per_step else d      # one full launch per step: the largest third
            out[k] = {"leg": leg, "launches_traced": len(d), "launches_used": len(top), "median_ns": top[len(top) // 2],
                      "min_ns": top[-1], "max_ns": top[0]}
    if out:
        p = os.path.join(ROOT, "profiles", f"{rnd}_full_launch.json")
        json.dump({"how": "rocprofv3 --kernel-trace of `python bench.py [--mode eval] --no-cpu-baseline --no-host-probe --no-kernel-events "
                          "--steps 20 --warmup 5`: per-launch End - Start; scorer: median over the largest launch of each step",
                   "kernels": out}, open(p, "w"), indent=1)
        print("wrote", p)


def readme(rnd):
    """profiles/README.md from the condensed files + the bench lines (gpurun_out/bench_<rnd>_{train,eval}.json)."""
    P = os.path.join(ROOT, "profiles")

    def table(f, n, steps):
        rows = list(csv.reader(open(f)))[1:]
        return "\n".join(f"| `{r[0][:70]}` | {int(r[1]) / steps:.0f} | {float(r[3]) / 1e3:.1f} | {float(r[2]) / steps / 1e6:.3f} | {r[4]} |"
                         for r in rows[:n] if r[3])

    for src, dst in ((f"bench_{rnd}_train.json", f"{rnd}_bench_train.json"), (f"bench_{rnd}_eval.json", f"{rnd}_bench_eval.json")):
        f = os.path.join(ROOT, "gpurun_out", src)
        if os.path.exists(f):
            json.dump(json.load(open(f)), open(os.path.join(P, dst), "w"), indent=1)
    bt = json.load(open(os.path.join(P, f"{rnd}_bench_train.json")))
    be = json.load(open(os.path.join(P, f"{rnd}_bench_eval.json")))
    t = json.load(open(os.path.join(P, f"{rnd}_hbm_traffic.json")))["kernels"]
    tr = "\n".join(f"| `{k}` | {v['hbm_read_bytes_per_launch'] / 1e6:.1f} | {v['hbm_write_bytes_per_launch'] / 1e6:.1f} |"
                   for k, v in t.items() if not k.startswith("__amd"))
    mu = ""
    for mode in ("", "_eval"):
        f = os.path.join(P, f"{rnd}_mfma_util{mode}.json")
        if os.path.exists(f):
            ks = json.load(open(f))["kernels"]
            mu += "\n".join(f"| `{k[:70]}` ({'eval' if mode else 'train'}) | {v['launches_sampled']} | {v['mfma_busy_cycles_per_launch']:.3g} | "
                            f"{v['shader_cycles_per_launch']:.3g} | {100 * v['mfma_util']:.1f} % | {100 * v['mfma_util_largest_launch']:.1f} % |"
                            for k, v in list(ks.items())[:10]) + "\n"
    steps = 25
    md = f"""# profiles/ - round {rnd[1:]} (one MI355X, gfx950, ROCm 7.2, cfg1 = HSTU Pixel8M shape)

All files here are condensed by `tools/summarize_profiles.py {rnd}` from rocprofv3 output written under `gpurun_out/` on the GPU box
by `tools/profile_round.sh {rnd}` (round 1's files, `r01_*`, are kept next to them for comparison).

| File | Command it comes from |
|---|---|
| `{rnd}_bench_train.json` | `python bench.py --steps 50 --warmup 10` - the bench line itself (train step replayed from a hipGraph; eval / gather / cpu_baseline sub-objects) |
| `{rnd}_bench_eval.json` | `python bench.py --mode eval --steps 20 --warmup 5` |
| `{rnd}_train_kernel_stats.csv` | `rocprofv3 --kernel-trace --stats --output-format csv -- python bench.py --no-cpu-baseline --no-host-probe --no-eval-leg --no-kernel-events --steps 20 --warmup 5` ({steps} steps traced) |
| `{rnd}_eval_kernel_stats.csv` | the same with `--mode eval` |
| `{rnd}_hbm_traffic.json`, `{rnd}_hbm_traffic_eval.json` | `rocprofv3 --pmc FETCH_SIZE -- python bench.py ... --steps 4 --warmup 4` and the same with `--pmc WRITE_SIZE` (separate passes); FETCH_SIZE KB x 2 (gfx950 correction), WRITE_SIZE KB x 1 |
| `{rnd}_mfma_util.json`, `{rnd}_mfma_util_eval.json` | `rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE -- python bench.py ...` (own pass); util = MFMA_BUSY / (1024 SIMDs x GRBM_GUI_ACTIVE / 8) |

## Train step (bench: {bt["value"]} seq/s, {bt["ms_per_step"]} ms/step; host issue {bt.get("host_enqueue_ms_per_step")} ms/step)

rocprofv3 kernel stats, per step ({steps} steps traced):

| kernel | launches/step | avg us | ms/step | % of GPU time |
|---|---|---|---|---|
{table(os.path.join(P, f"{rnd}_train_kernel_stats.csv"), 26, steps)}

HIP-event durations of the library's kernels from the bench line's evented pass (ms/step): `{json.dumps(bt["kernel_ms_per_step"])}`.

Roofline object of the bench line (dominant kernel of the step): `{json.dumps(bt.get("roofline"))}`.

Embedding gather: `{json.dumps(bt.get("gather"))}`.

Sampled softmax (executed rates: two MFMA products per distinct query row each way): `{json.dumps(bt.get("sampled_softmax"))}`.

## Eval step (bench: {be["value"]} users/s, {be["ms_per_step"]} ms/step of 256 users x 4 heads over 453 938 items, top-k ranked on fp32 scores)

| kernel | launches/step | avg us | ms/step | % of GPU time |
|---|---|---|---|---|
{table(os.path.join(P, f"{rnd}_eval_kernel_stats.csv"), 12, steps)}

Catalog scorer roofline (bench line): `{json.dumps(be.get("roofline"))}`.

## MFMA utilisation (PMC)

| kernel | launches | MFMA busy cycles / launch | shader cycles / launch | MFMA util | util of the largest launch |
|---|---|---|---|---|---|
{mu}
## HBM traffic per launch (PMC, MB; train leg)

| kernel | read | written |
|---|---|---|
{tr}
"""
    open(os.path.join(P, "README.md"), "w").write(md)
    print("wrote", os.path.join(P, "README.md"))


if __name__ == "__main__":
    rnd = sys.argv[1] if len(sys.argv) > 1 else "r01"
    os.makedirs(os.path.join(ROOT, "profiles"), exist_ok=True)
    for leg in ("train", "eval", "hllm_train"):
        kernel_stats(rnd, leg)
    traffic(rnd)
    traffic(rnd, "eval_")
    mfma_util(rnd)
    mfma_util(rnd, "eval_")
    full_launches(rnd)
    for extra in ("emu", "dp2_gloo", "dp1_rccl"):                       # bench --emulate-world / the two-rank gloo rehearsal lines, if the set made them
        f = os.path.join(ROOT, "gpurun_out", f"bench_{rnd}_{extra}.json")
        if os.path.exists(f) and os.path.getsize(f):
            json.dump(json.loads(open(f).read().strip().splitlines()[-1]), open(os.path.join(ROOT, "profiles", f"{rnd}_bench_{extra}.json"), "w"), indent=1)
    readme(rnd)
