#!/bin/bash
# per-kernel statistics of one training step at adim 512 / aheads 8 (tools/bench_width.py).  usage: bash tools/profile_width.sh <tag> [fp32|bf16]
set -u
tag=${1:-w512}; prec=${2:-fp32}
root=$(pwd)
cd /tmp && export TMPDIR=/tmp
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d "$root/gpurun_out/pw_$tag" -- python3 "$root/tools/bench_width.py" --precision $prec --steps 20 ${EAMD_WIDTH_ARGS:-} > "$root/gpurun_out/pw_$tag.log" 2>&1 || { tail -n 5 "$root/gpurun_out/pw_$tag.log"; exit 1; }
cd "$root"
grep "ms per step" gpurun_out/pw_$tag.log
f=$(find gpurun_out/pw_$tag -name "*kernel_stats.csv" | head -1)
cp "$f" gpurun_out/${tag}_kernel_stats.csv
rm -rf gpurun_out/pw_$tag
python tools/kstats.py gpurun_out/${tag}_kernel_stats.csv 32
