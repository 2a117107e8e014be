#!/bin/bash
# PMC pass over the attention kernels alone (tools/attn_pmc_driver.py: config 2's shapes, fp32): where do the waves of
# attn_f32_fwd_kernel / attn_f32_bwd_q_kernel / attn_f32_bwd_kv_kernel spend their cycles.  usage: bash tools/pmc_attn.sh <tag>
set -u
tag=${1:-attn}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for C in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR"; do
  t=$(echo $C | cut -d' ' -f1)
  rm -rf gpurun_out/pmca_$t
  timeout -k 10 300 rocprofv3 --pmc $C --output-format csv -d gpurun_out/pmca_$t -- python tools/attn_pmc_driver.py > gpurun_out/pmca_$t.log 2>&1 || { tail -n 3 gpurun_out/pmca_$t.log; exit 1; }
done
python - <<PY
import csv, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.Counter()
for f in glob.glob("gpurun_out/pmca_*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][:44]
        if "attn" not in k:
            continue
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
        cnt[(k, r["Counter_Name"])] += 1
for k, d in sorted(agg.items()):
    n = max(1, cnt[(k, "SQ_WAVE_CYCLES")])
    wc = d.get("SQ_WAVE_CYCLES", 1.0)
    print(k, "launches", n)
    for c in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_WAIT_INST_LDS", "SQ_ACTIVE_INST_LDS", "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_VMEM"):
        if c in d:
            print("   %-22s %6.3f of wave cycles" % (c, d[c] / wc))
    if "SQ_VALU_MFMA_BUSY_CYCLES" in d and "SQ_BUSY_CYCLES" in d:
        print("   MFMA busy / SQ busy    %6.3f" % (d["SQ_VALU_MFMA_BUSY_CYCLES"] / d["SQ_BUSY_CYCLES"]))
    if "SQ_LDS_BANK_CONFLICT" in d:
        print("   LDS bank conflict / LDS active %6.3f" % (d["SQ_LDS_BANK_CONFLICT"] / max(1.0, d["SQ_LDS_IDX_ACTIVE"])))
    for c in ("SQ_INSTS_VALU", "SQ_INSTS_LDS", "SQ_INSTS_VMEM_RD", "SQ_INSTS_VMEM_WR"):
        if c in d:
            print("   %-22s %10.0f per launch" % (c, d[c] / max(1, cnt[(k, c)])))
PY
rm -rf gpurun_out/pmca_*
