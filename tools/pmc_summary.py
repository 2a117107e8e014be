#!/usr/bin/env python3
"""Condense rocprofv3 --pmc passes (tools/pmc_gemm.sh) into profiles/: per-kernel counter sums and the
HBM-traffic figure bench.py reports as roofline.traffic.

FETCH_SIZE / WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE counts 128-B requests as 64 B, so it is doubled
(MI355X_MICROARCH.md, section HBM).  usage: pmc_summary.py <gpurun_out dir> <profiles dir> <tag> <gemm launches/step> [precision]
"""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

src, dst, tag, lps = sys.argv[1], sys.argv[2], sys.argv[3], float(sys.argv[4])
prec = sys.argv[5] if len(sys.argv) > 5 else None
pmcdir = tag + "_pmc" + ("_" + prec if prec else "")
os.makedirs(os.path.join(dst, pmcdir), exist_ok=True)
tot = {}
for d in sorted(glob.glob(os.path.join(src, "pmc_*"))):
    if not os.path.isdir(d):
        continue
    files = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
    if not files:
        continue
    agg = defaultdict(lambda: [0, 0.0])
    with open(files[0]) as f:
        for r in csv.DictReader(f):
            a = agg[(r["Kernel_Name"], r["Counter_Name"])]
            a[0] += 1
            a[1] += float(r["Counter_Value"])
    name = os.path.basename(d)[4:]
    with open(os.path.join(dst, pmcdir, name + "_summary.csv"), "w") as f:
        f.write("kernel,counter,dispatches,sum,mean\n")
        for (k, c), (n, s) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
            f.write('"%s",%s,%d,%r,%r\n' % (k, c, n, s, s / n))
    for (k, c), (n, s) in agg.items():
        t = tot.setdefault(c, dict(gemm=[0, 0.0], all=[0, 0.0]))
        t["all"][0] += n; t["all"][1] += s
        # the MFMA-contraction family: GEMM kernels, fused FFN, row-block projections, fused attention (not the decode kernels)
        if ("gemm" in k or "attn_" in k or ("ffn_" in k and "pack" not in k) or "rowproj_f32_kernel" in k) and "decode_" not in k:
            t["gemm"][0] += n; t["gemm"][1] += s
if "FETCH_SIZE" in tot and "WRITE_SIZE" in tot:
    steps = tot["FETCH_SIZE"]["gemm"][0] / lps
    if abs(steps - round(steps)) < 0.03 * steps:     # a few family launches outside the step (set-up GEMMs): whole steps
        steps = float(round(steps))
    fg = tot["FETCH_SIZE"]["gemm"][1] * 1024 * 2 / steps
    wg = tot["WRITE_SIZE"]["gemm"][1] * 1024 / steps
    out = {
        "source": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes, tools/pmc_gemm.sh), bench.py --no-graph; "
                  "FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950 tallies 128-B requests at 64 B)",
        "steps_profiled": steps,
        "gemm_family": {"launches_per_step": lps, "fetch_bytes_per_step": fg, "write_bytes_per_step": wg,
                        "hbm_bytes_per_launch": (fg + wg) / lps},
        "all_kernels": {"fetch_bytes_per_step": tot["FETCH_SIZE"]["all"][1] * 2048 / steps,
                        "write_bytes_per_step": tot["WRITE_SIZE"]["all"][1] * 1024 / steps},
    }
    for c, t in tot.items():
        if c not in ("FETCH_SIZE", "WRITE_SIZE"):
            out.setdefault("sq_counters_gemm_family_sum", {})[c] = t["gemm"][1]
    with open(os.path.join(dst, tag + "_pmc_traffic" + ("_" + prec if prec else "") + ".json"), "w") as f:
        json.dump(out, f, indent=1)
    print(json.dumps(out, indent=1))
