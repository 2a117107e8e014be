#!/usr/bin/env python3
"""From a rocprofv3 kernel_trace.csv: kernels of the LAST complete training step (between the last two adam_kernel
launches), grouped by name with count and total time; gaps between consecutive kernels summed.
usage: tools/trace_step.py <kernel_trace.csv> [n]"""
import collections, csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
adam = [i for i, r in enumerate(rows) if r["Kernel_Name"].startswith("adam_kernel") or "adam_kernel" in r["Kernel_Name"]]
lo, hi = adam[-2] + 1, adam[-1] + 1
step = rows[lo:hi]
t0, t1 = int(step[0]["Start_Timestamp"]), int(step[-1]["End_Timestamp"])
busy = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in step)
agg = collections.defaultdict(lambda: [0, 0])
for r in step:
    n = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "")[:90]
    agg[n][0] += 1
    agg[n][1] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
print("step: %d kernels, wall %.3f ms, kernel-busy %.3f ms, idle %.3f ms" % (len(step), (t1 - t0) / 1e6, busy / 1e6, (t1 - t0 - busy) / 1e6))
for n, (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[: int(sys.argv[2]) if len(sys.argv) > 2 else 60]:
    print("%5d %9.1f us %8.2f us each  %s" % (c, t / 1e3, t / 1e3 / c, n))
