"""Where a replayed train step's wall time goes between kernels: from a rocprofv3 --kernel-trace CSV of `bench.py`, the last
steps' busy time (union of kernel intervals), idle gaps, and the largest gaps with the kernels on either side.
python tools/step_timeline.py <kernel_trace.csv> [n_last_steps]"""
import csv, sys, collections
f = sys.argv[1]
n_last = int(sys.argv[2]) if len(sys.argv) > 2 else 10
rows = []
for r in csv.DictReader(open(f)):
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id", "0")))
rows.sort()
# a step = from one gather_step kernel (first kernel of the forward) to the next
marks = [i for i, r in enumerate(rows) if "gather_step_kernel" in r[2]]
steps = [(marks[i], marks[i + 1]) for i in range(len(marks) - 1)][-n_last:]
tot_wall = tot_busy = 0
gaps = collections.Counter()
for a, b in steps:
    seg = rows[a:b]
    wall = rows[b][0] - seg[0][0]
    cur_end = seg[0][0]
    busy = 0
    last = None
    for s, e, name, q in seg:
        if s > cur_end:
            gaps[(last[:40] if last else "", name[:40])] += s - cur_end
            cur_end = s
        if e > cur_end:
            busy += e - max(s, cur_end)
            cur_end = e
            last = name
    if rows[b][0] > cur_end:
        gaps[(last[:40], "next step's " + rows[b][2][:28])] += rows[b][0] - cur_end
    tot_wall += wall
    tot_busy += busy
n = len(steps)
print(f"{n} steps: wall {tot_wall / n / 1e6:.3f} ms per step, GPU busy (union of kernel intervals) {tot_busy / n / 1e6:.3f} ms, idle {(tot_wall - tot_busy) / n / 1e6:.3f} ms")
print("largest idle gaps (us per step), between the kernel that ended last and the kernel that started next:")
for (a_, b_), v in gaps.most_common(25):
    print(f"  {v / n / 1e3:7.1f}  {a_:40s} -> {b_}")
