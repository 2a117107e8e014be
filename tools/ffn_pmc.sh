#!/bin/bash
# PMC passes over tools/ffn_pmc.py (separate passes; no trace domains beside --pmc)
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
i=0
for C in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_INSTS_VALU" "GRBM_GUI_ACTIVE TCC_HIT_sum TCC_MISS_sum" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_LDS SQ_INSTS_VMEM"; do
  i=$((i+1))
  timeout -k 10 200 rocprofv3 --pmc $C --output-format csv -d gpurun_out/ffnpmc_$i -- python tools/ffn_pmc.py 4 > gpurun_out/ffnpmc_$i.log 2>&1 || { echo "pass $i failed"; tail -5 gpurun_out/ffnpmc_$i.log; }
done
python - <<'PY'
import csv, glob, collections
for d in sorted(glob.glob("gpurun_out/ffnpmc_*")):
    fs = glob.glob(d + "/**/*counter_collection.csv", recursive=True)
    if not fs: continue
    agg = collections.defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open(fs[0])):
        k = r["Kernel_Name"]
        if "ffn_f32" in k or "gemm_f32" in k:
            a = agg[(k[:90], r["Counter_Name"])]; a[0] += 1; a[1] += float(r["Counter_Value"])
    for (k, c), (n, s) in sorted(agg.items()):
        print("%-92s %-28s n=%d mean=%.4g" % (k, c, n, s / n))
PY
