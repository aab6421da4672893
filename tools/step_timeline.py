"""Per-step kernel timeline from a rocprofv3 --kernel-trace CSV: start offset, duration, gap to the previous kernel's end, name.
usage: python tools/step_timeline.py <kernel_trace.csv> [out.txt]     (the last full step between two adam_flat_kernel launches)"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if r["Kernel_Name"].startswith("adam_flat_kernel")]
a, b = idx[-2], idx[-1]
out = open(sys.argv[2], "w") if len(sys.argv) > 2 else sys.stdout
t0 = int(rows[a + 1]["Start_Timestamp"])
end_max = None
busy = 0
for r in rows[a + 1:b + 1]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    gap = (s - end_max) if end_max is not None else 0
    busy += e - s
    out.write("%8.1f %7.1f gap %6.1f  q%s  %s\n" % ((s - t0) / 1e3, (e - s) / 1e3, gap / 1e3, r.get("Queue_Id", "?"), r["Kernel_Name"][:90]))
    end_max = e if end_max is None else max(end_max, e)
span = (int(rows[b]["End_Timestamp"]) - t0) / 1e3
print("%d kernels, span %.1f us, sum of durations %.1f us" % (b - a, span, busy / 1e3), file=sys.stderr)
