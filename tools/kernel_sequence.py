#!/usr/bin/env python3
"""Kernels (name, start offset, duration) between two consecutive launches of a marker kernel in a rocprofv3 kernel
trace - e.g. one decoder step of the RNN model: kernel_sequence.py <rocprof output dir> <kernel-name substring>"""
import csv, sys, glob
f = glob.glob(sys.argv[1] + '/**/*kernel_trace.csv', recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
names = [r['Kernel_Name'] for r in rows]
key = sys.argv[2]
idx = [i for i, n in enumerate(names) if key in n]
# take a window in the middle of the last replay
i0, i1 = idx[-40], idx[-39]
t0 = int(rows[i0]['Start_Timestamp'])
print("kernels between two", key, ":", i1 - i0, "span us", (int(rows[i1]['Start_Timestamp']) - t0) / 1e3)
for r in rows[i0:i1]:
    n = r['Kernel_Name'].replace('void (anonymous namespace)::', '').replace('(anonymous namespace)::', '')
    print("%8.1f  %6.1f us  %s" % ((int(r['Start_Timestamp']) - t0) / 1e3, (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3, n[:110]))
