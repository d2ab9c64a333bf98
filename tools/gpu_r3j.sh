#!/bin/bash
set -u
TAG=${1:-r03j}
OUT=gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
for v in "--batch 8 --in-flight 2" "--batch 8 --in-flight 2" "--batch 10 --in-flight 2" "--batch 20 --in-flight 1" "--batch 8 --in-flight 3" "--batch 7 --in-flight 3"; do
  echo "== --steps 20 --warmup 5 $v" | tee -a $OUT/bench_debug.txt
  BENCH_DEBUG=1 timeout 120 python bench.py --steps 20 --warmup 5 --balance 0 $v --no-extras 2>&1 | grep "enqueue\|main line" | tee -a $OUT/bench_debug.txt
done
timeout 600 python -m pytest tests/test_gpu_model.py -q --no-header -p no:cacheprovider -s -k "cfg1" > $OUT/pytest_cfg1.log 2>&1; echo "cfg1 exit $?"; grep "cfg1\|passed\|failed\|^E " $OUT/pytest_cfg1.log | tail
exit 0
