#!/usr/bin/env python3
"""print the top rows of a rocprofv3 kernel_stats.csv compactly: tools/kstats.py <csv> [n] [steps]"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
n = int(sys.argv[2]) if len(sys.argv) > 2 else 25
tot = sum(float(r["TotalDurationNs"]) for r in rows)
for r in rows[:n]:
    name = r["Name"].replace("(anonymous namespace)::", "").replace("void ", "")[:72]
    print(f"{name:72s} {int(r['Calls']):6d} {float(r['AverageNs']) / 1e3:9.1f} us {float(r['TotalDurationNs']) / tot * 100:5.2f}%")
print("total kernel time %.1f ms" % (tot / 1e6))
