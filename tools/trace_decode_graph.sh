#!/bin/bash
# rocprofv3 kernel trace of graph-replayed single-utterance beam steps (tools/decode_graph_trace.py): per (kernel, grid) average
# durations, and the busy / idle split of a replayed step's timeline.   usage (GPU box, repo root): bash tools/trace_decode_graph.sh <tag>
set -u
tag=${1:-x}
root=$(pwd)
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --output-format csv -d "$root/gpurun_out/gprof_$tag" -- \
  python3 "$root/tools/decode_graph_trace.py" > "$root/gpurun_out/gprof_$tag.log" 2>&1 || { tail -n 5 "$root/gpurun_out/gprof_$tag.log"; exit 1; }
cd "$root"
grep "search" gpurun_out/gprof_$tag.log
f=$(find gpurun_out/gprof_$tag -name "*kernel_trace.csv" | head -1)
python tools/step_timeline.py "$f" > gpurun_out/${tag}_decode_step_timeline.txt 2>&1
gzip -c "$f" > gpurun_out/${tag}_kernel_trace.csv.gz
rm -rf gpurun_out/gprof_$tag
cat gpurun_out/${tag}_decode_step_timeline.txt
