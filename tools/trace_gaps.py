"""Busy / idle summary of a rocprofv3 --kernel-trace CSV over its last `frac` of kernels: wall span, union of kernel intervals, the
largest kernels by total time.  usage: python tools/trace_gaps.py <kernel_trace.csv> [frac=0.5]"""
import collections
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
frac = float(sys.argv[2]) if len(sys.argv) > 2 else 0.5
rows = rows[int(len(rows) * (1 - frac)):]
t0, t1 = int(rows[0]["Start_Timestamp"]), max(int(r["End_Timestamp"]) for r in rows)
busy, cur_e = 0, t0
tot = collections.defaultdict(lambda: [0, 0])
for r in rows:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    if e > cur_e:
        busy += e - max(s, cur_e)
        cur_e = e
    k = r["Kernel_Name"][:70]
    tot[k][0] += 1
    tot[k][1] += e - s
print("kernels %d  span %.1f us  busy %.1f us (%.0f%%)" % (len(rows), (t1 - t0) / 1e3, busy / 1e3, 100.0 * busy / (t1 - t0)))
for k, (n, d) in sorted(tot.items(), key=lambda kv: -kv[1][1])[:25]:
    print("%9.1f us  %5d x  %s" % (d / 1e3, n, k))
