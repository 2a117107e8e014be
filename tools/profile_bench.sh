#!/bin/bash
# rocprofv3 kernel trace of the default bench (hipGraph launch) -> gpurun_out/prof_<tag>/ + a per-kernel csv summary.
# usage (on the GPU box, from the repo root): bash tools/profile_bench.sh <tag> [bench args]
set -u
tag=${1:-x}; shift || true
root=$(pwd)
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d "$root/gpurun_out/prof_$tag" -- \
  python3 "$root/bench.py" --steps 20 --warmup 3 --no-cpu-baseline --no-extra-configs --no-decode "$@" > "$root/gpurun_out/prof_$tag.log" 2>&1 || exit 1
cd "$root"
f=$(find gpurun_out/prof_$tag -name "*kernel_stats.csv" | head -1)
echo "stats: $f"
head -25 "$f" | cut -c1-160
