"""a rocprofv3 kernel trace (csv) of tools/decode_graph_trace.py -> the last search's steps: per (kernel, grid) count per step and
average duration, the sum of kernel time per step, and the step's span (first start .. last end) = busy + gaps"""
import csv, sys, collections
import gzip
rows = list(csv.DictReader(gzip.open(sys.argv[1], "rt") if sys.argv[1].endswith(".gz") else open(sys.argv[1])))
def short(nm):
    nm = nm.replace("(anonymous namespace)::", "").replace("void ", "").replace("at::native::", "")
    return nm.split("(")[0][:46]
ev = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], int(r["Grid_Size_X"]) // max(1, int(r["Workgroup_Size_X"])))
             for r in rows), key=lambda e: e[0])
# steps of a search start with the embedding kernel; the last search = the last run of consecutive steps
starts = [i for i, e in enumerate(ev) if "embed_pe_kernel" in e[2]]
if len(starts) < 4:
    print("events %d, embed starts %d; names: %s" % (len(ev), len(starts), sorted({e[2][:40] for e in ev})[:30]))
nstep = int(sys.argv[2]) if len(sys.argv) > 2 else 24
first = starts[-nstep]
seg = ev[first:]
bounds = starts[-nstep:] + [len(ev)]
spans, busys, counts = [], [], []
per = collections.defaultdict(lambda: [0, 0])
for a, b in zip(bounds[:-1], bounds[1:]):
    st = ev[a:b]
    if b == len(ev):          # the tail after the last step's kernels (log read-back etc.) is not a step
        st = st[:max(1, len(ev[bounds[-3]:bounds[-2]]))]
    spans.append(max(e[1] for e in st) - st[0][0]); busys.append(sum(e[1] - e[0] for e in st)); counts.append(len(st))
    for e in st:
        p = per[(short(e[2]), e[3])]; p[0] += 1; p[1] += e[1] - e[0]
n = len(spans)
# step-to-step period: start of step i+1 - start of step i
periods = [ev[bounds[i + 1]][0] - ev[bounds[i]][0] for i in range(n - 1)]
print("periods (us):", " ".join("%.0f" % (p / 1e3) for p in periods))
print("steps %d | kernels per step %.1f | period %.1f us | span %.1f us | kernel time (sum, overlapping streams counted twice) %.1f us" %
      (n, sum(counts) / n, sum(periods) / len(periods) / 1e3, sum(spans) / n / 1e3, sum(busys) / n / 1e3))
for (name, grid), (c, t) in sorted(per.items(), key=lambda kv: -kv[1][1]):
    print("%6.2f x %7.2f us = %7.1f us/step  %-46s grid %d" % (c / n, t / c / 1e3, t / n / 1e3, name, grid))
