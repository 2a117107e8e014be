#!/bin/bash
# sweep of the grouped weight-gradient knobs (fp32 and bf16 step time): "T128_MIN T128_WGS"
for cfg in "100 128" "100 64" "100 96" "100 32"; do
  set -- $cfg
  for prec in fp32; do
    EAMD_GROUP_T128_MIN=$1 EAMD_GROUP_T128_WGS=$2 timeout -k 10 200 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-hbm-roofline --no-second-precision --precision $prec 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('t128min $1 wgs $2 $prec: %.3f ms, gemm %.3f ms, launches %d' % (d['ms_per_step'], d['roofline']['gemm_ms_per_step'], d['roofline']['launches_per_step']))
" || exit 1
  done
done
