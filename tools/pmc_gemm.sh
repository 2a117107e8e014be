#!/bin/bash
# PMC passes for the dominant kernel (separate passes: TCC slots cannot hold FETCH_SIZE and WRITE_SIZE together)
set -u
P=${1:-fp32}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for C in "FETCH_SIZE" "WRITE_SIZE" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES"; do
  tag=$(echo $C | cut -d' ' -f1)
  timeout -k 10 400 rocprofv3 --pmc $C --output-format csv -d gpurun_out/pmc_$tag -- python bench.py --steps 2 --warmup 1 --no-graph --no-cpu-baseline --no-second-precision --no-hbm-roofline --no-extra-configs --no-decode --precision $P > gpurun_out/pmc_$tag.log 2>&1 || exit 1
done
ls -R gpurun_out/pmc_* | head -30
