cd /tmp && export TMPDIR=/tmp
root=$GRAFT_REPO_ROOT
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $root/gpurun_out/tr_decg -- python3 $root/tools/bench_decode_graph.py 2 10 0.3 0.2 > $root/gpurun_out/tr_decg.log 2>&1 || exit 1
cd $root
f=$(find gpurun_out/tr_decg -name "*kernel_trace.csv" | head -1)
python - "$f" <<'PY' > gpurun_out/tr_decg_summary.txt
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
# the last 98 steps are the eager reference; before them the timed graph replays: take the window of the timed replays by
# locating the last two thirds... simpler: summarise the LAST 40% of the graph-phase kernels = kernels between 45% and 70% of the run
n = len(rows)
seg = rows[int(n * 0.55): int(n * 0.72)]
c = collections.Counter(); t = collections.Counter()
for r in seg:
    k = r['Kernel_Name'].replace('void (anonymous namespace)::', '').replace('(anonymous namespace)::', '')[:90]
    c[k] += 1; t[k] += (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
span = (int(seg[-1]['End_Timestamp']) - int(seg[0]['Start_Timestamp'])) / 1e3
print(len(seg), "kernels, span %.1f us, busy %.1f us" % (span, sum(t.values())))
for k, v in sorted(t.items(), key=lambda kv: -kv[1])[:28]:
    print("%6d %9.1f us %7.2f each  %s" % (c[k], v, v / c[k], k))
PY
rm -rf gpurun_out/tr_decg
