#!/bin/bash
# One round's judged evidence: kernel stats of the default bench (both precisions in one run) + the PMC traffic passes
# of each precision, condensed into profiles/<tag>_*.  usage (on the GPU box): bash tools/profile_round.sh <tag> <family launches/step fp32> <.. bf16>
set -u
tag=$1; lf=$2; lb=$3
bash tools/profile_bench.sh $tag || exit 1
f=$(find gpurun_out/prof_$tag -name "*kernel_stats.csv" | head -1)
cp "$f" gpurun_out/${tag}_kernel_stats.csv
grep '^{' gpurun_out/prof_$tag.log > gpurun_out/${tag}_bench.json
rm -rf gpurun_out/prof_$tag
for P in fp32 bf16; do
  rm -rf gpurun_out/pmc_*
  bash tools/pmc_gemm.sh $P > /dev/null 2>&1 || exit 1
  l=$lf; [ $P = bf16 ] && l=$lb
  mkdir -p gpurun_out/profiles_$tag
  python tools/pmc_summary.py gpurun_out gpurun_out/profiles_$tag $tag $l $P > /dev/null || exit 1
done
rm -rf gpurun_out/pmc_*
ls gpurun_out/profiles_$tag
