#!/bin/bash
set -u
TAG=${1:-r03k}
OUT=gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
for v in "--batch 8 --in-flight 1" "--batch 10 --in-flight 1" "--batch 16 --in-flight 1" "--batch 12 --in-flight 1" "--batch 7 --in-flight 1" "--batch 8 --in-flight 2" "--batch 10 --in-flight 1" "--batch 8 --in-flight 1"; do
  echo "== --steps 20 --warmup 5 $v" | tee -a $OUT/bench_debug.txt
  BENCH_DEBUG=1 timeout 120 python bench.py --steps 20 --warmup 5 --balance 0 $v --no-extras 2>&1 | grep "enqueue\|main line" | tee -a $OUT/bench_debug.txt
done
for v in "--batch 10 --in-flight 1" "--batch 16 --in-flight 1" "--batch 10 --in-flight 2"; do
  echo "== --steps 240 --warmup 24 $v" | tee -a $OUT/bench_debug.txt
  timeout 120 python bench.py --steps 240 --warmup 24 --balance 0 $v --no-extras 2>&1 | grep "main line" | tee -a $OUT/bench_debug.txt
done
exit 0
