#!/bin/bash
# rocprofv3 kernel trace of single-utterance beam searches (tools/decode_launch_census.py) -> per-kernel average durations
# usage (GPU box, repo root): bash tools/trace_decode_step.sh <tag>
set -u
tag=${1:-x}
root=$(pwd)
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$root/gpurun_out/dprof_$tag" -- \
  python3 "$root/tools/decode_launch_census.py" > "$root/gpurun_out/dprof_$tag.log" 2>&1 || exit 1
cd "$root"
f=$(find gpurun_out/dprof_$tag -name "*kernel_stats.csv" | head -1)
cp "$f" gpurun_out/${tag}_decode_kernel_stats.csv
rm -rf gpurun_out/dprof_$tag
python tools/kstats.py gpurun_out/${tag}_decode_kernel_stats.csv 30
