cd /tmp && export TMPDIR=/tmp
root=$GRAFT_REPO_ROOT
SGD_RATIO=0.2 SGD_NB=32 timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $root/gpurun_out/tr_decb -- python3 $root/tools/step_graph_debug.py batch > $root/gpurun_out/tr_decb.log 2>&1 || exit 1
cd $root
f=$(find gpurun_out/tr_decb -name "*kernel_trace.csv" | head -1)
python - "$f" <<'PY' > gpurun_out/tr_decb_summary.txt
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
n = len(rows)
seg = rows[int(n * 0.86):]          # the last (replaying) batched search
c = collections.Counter(); t = collections.Counter()
for r in seg:
    k = r['Kernel_Name'].replace('void (anonymous namespace)::', '').replace('(anonymous namespace)::', '')[:90]
    c[k] += 1; t[k] += (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
span = (int(seg[-1]['End_Timestamp']) - int(seg[0]['Start_Timestamp'])) / 1e3
print(len(seg), "kernels, span %.1f us, busy %.1f us" % (span, sum(t.values())))
for k, v in sorted(t.items(), key=lambda kv: -kv[1])[:22]:
    print("%6d %9.1f us %7.2f each  %s" % (c[k], v, v / c[k], k))
PY
rm -rf gpurun_out/tr_decb
