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


def traffic(rnd):
    acc = {}
    for leg, counter, scale in (("fetch", "FETCH_SIZE", 2.0 * 1024), ("write", "WRITE_SIZE", 1024.0)):
        f = newest(os.path.join(ROOT, "gpurun_out", f"pmc_{rnd}_{leg}", "*", "*_counter_collection.csv"))
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
                  "hbm_bytes_per_launch": round(fb + wb)}
    if out:
        p = os.path.join(ROOT, "profiles", f"{rnd}_hbm_traffic.json")
        json.dump({"how": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes of `python bench.py --no-cpu-baseline "
                          "--steps 4 --warmup 2`; FETCH_SIZE KB x 2 (gfx950 tallies 128-B requests at 64 B), WRITE_SIZE KB x 1; "
                          "mean over all launches of the kernel", "kernels": out}, open(p, "w"), indent=1)
        print("wrote", p)


if __name__ == "__main__":
    rnd = sys.argv[1] if len(sys.argv) > 1 else "r01"
    os.makedirs(os.path.join(ROOT, "profiles"), exist_ok=True)
    for leg in ("train", "eval"):
        kernel_stats(rnd, leg)
    traffic(rnd)
