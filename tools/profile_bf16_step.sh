set -u
root=$(pwd)
cd /tmp && export TMPDIR=/tmp
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d "$root/gpurun_out/pbf16" -- python3 "$root/bench.py" --steps 20 --warmup 3 --precision bf16 --no-second-precision --no-cpu-baseline --no-hbm-roofline --no-decode --no-extra-configs > "$root/gpurun_out/pbf16.log" 2>&1 || { tail -n 5 "$root/gpurun_out/pbf16.log"; exit 1; }
cd "$root"
tail -n 1 gpurun_out/pbf16.log | cut -c1-300
f=$(find gpurun_out/pbf16 -name "*kernel_stats.csv" | head -1)
cp "$f" gpurun_out/r5_bf16_only_kernel_stats.csv
rm -rf gpurun_out/pbf16
python tools/kstats.py gpurun_out/r5_bf16_only_kernel_stats.csv 45
