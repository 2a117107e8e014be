#!/bin/bash
# Runs the GPU test tiers one process per file; stops at the first tier that dies abnormally
# (signal / timeout) so a faulting kernel is never followed by more GPU work in the same call.
set -u
mkdir -p gpurun_out
run() {
  name=$1; shift
  echo "=== $name: $*" | tee -a gpurun_out/ci.log
  timeout -k 10 "${TIER_TIMEOUT:-420}" "$@" > "gpurun_out/$name.log" 2>&1
  rc=$?
  echo "=== $name rc=$rc" | tee -a gpurun_out/ci.log
  tail -n 40 "gpurun_out/$name.log" | grep -E "passed|failed|error|rc=|Error|FAILED" | tail -n 25
  if [ $rc -ne 0 ] && [ $rc -ne 1 ]; then echo "abnormal exit, stopping"; exit $rc; fi
  return 0
}
: > gpurun_out/ci.log
for t in "$@"; do
  case $t in
    ops)   run test_ops python -m pytest tests/test_gpu_ops.py -m gpu -q -s --tb=short -p no:cacheprovider ;;
    model) run test_model python -m pytest tests/test_gpu_model.py -m gpu -q -s --tb=short -p no:cacheprovider ;;
    rnn)   run test_rnn python -m pytest tests/test_gpu_rnn.py -m gpu -q -s --tb=short -p no:cacheprovider ;;
    smoke) run smoke python -c "import __graft_entry__ as g; g.smoke()" ;;
    bench) run bench python bench.py --steps 5 --warmup 2 ;;
    *) echo "unknown tier $t"; exit 2 ;;
  esac
done
exit 0
