#!/bin/bash
# kernel trace of a short bench run -> per-kernel summary of the last complete step: bash tools/trace_bench.sh <tag> [bench args]
tag=${1:-x}; shift || true
root=$(pwd)
cd /tmp && export TMPDIR=/tmp
timeout -k 10 500 rocprofv3 --kernel-trace --output-format csv -d $root/gpurun_out/tr_$tag -- python3 $root/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-second-precision --no-hbm-roofline "$@" > $root/gpurun_out/tr_$tag.log 2>&1 || exit 1
cd $root
f=$(find gpurun_out/tr_$tag -name "*kernel_trace.csv" | head -1)
python tools/trace_step.py $f 70 > gpurun_out/tr_step_$tag.txt
rm -rf gpurun_out/tr_$tag
