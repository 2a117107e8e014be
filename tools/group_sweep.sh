#!/bin/bash
# sweep of the grouped weight-gradient knobs (fp32 and bf16 step time)
for cfg in "192 1" "384 1" "384 2" "1024 1" "1024 2"; do
  set -- $cfg
  for prec in fp32 bf16; do
    EAMD_GROUP_MAX_TILES=$1 EAMD_GROUP_SK_DIV=$2 timeout -k 10 200 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-hbm-roofline --no-second-precision --precision $prec 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('tiles $1 skdiv $2 $prec: %.3f ms, gemm %.3f ms, launches %d' % (d['ms_per_step'], d['roofline']['gemm_ms_per_step'], d['roofline']['launches_per_step']))
" || exit 1
  done
done
