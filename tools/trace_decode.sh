cd /tmp && export TMPDIR=/tmp
root=$GRAFT_REPO_ROOT
timeout -k 10 400 rocprofv3 --kernel-trace --output-format csv -d $root/gpurun_out/tr_dec -- python3 $root/tools/bench_decode.py --utts 1 > $root/gpurun_out/tr_dec.log 2>&1 || exit 1
cd $root
f=$(find gpurun_out/tr_dec -name "*kernel_trace.csv" | head -1)
python - "$f" <<'PY' > gpurun_out/tr_dec_summary.txt
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
c = collections.Counter(); t = collections.Counter()
for r in rows:
    n = r['Kernel_Name'].replace('void (anonymous namespace)::', '').replace('(anonymous namespace)::', '')[:100]
    c[n] += 1; t[n] += (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
print(len(rows), "kernels")
for n, k in c.most_common(70):
    print("%7d %10.1f us %6.2f us each  %s" % (k, t[n], t[n] / k, n))
PY
rm -rf gpurun_out/tr_dec
