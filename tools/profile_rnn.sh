#!/bin/bash
# rocprofv3 kernel stats of tools/bench_rnn.py for one config + precision -> gpurun_out/<tag>_kernel_stats.csv
# usage: bash tools/profile_rnn.sh <tag> <4|5> <fp32|bf16>
set -u
tag=$1; which=$2; prec=$3
root=$(pwd)
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d "$root/gpurun_out/prof_$tag" -- \
  python3 "$root/tools/bench_rnn.py" --which $which --precision $prec > "$root/gpurun_out/prof_$tag.log" 2>&1 || exit 1
cd "$root"
f=$(find gpurun_out/prof_$tag -name "*kernel_stats.csv" | head -1)
cp "$f" gpurun_out/${tag}_kernel_stats.csv
grep "^config" gpurun_out/prof_$tag.log | cut -c1-200
rm -rf gpurun_out/prof_$tag
head -30 gpurun_out/${tag}_kernel_stats.csv | cut -c1-170
