#!/bin/bash
set -u
TAG=${1:-r03r}
OUT=gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 1500 python -m pytest tests/test_gpu_train.py tests/test_gpu_bench.py -q --no-header -p no:cacheprovider > $OUT/pytest_train.log 2>&1; echo "train+bench tests exit $?"; tail -5 $OUT/pytest_train.log
for m in f32 bf16 f32_mfma; do
  timeout 300 python bench.py --workload train --train-dtype $m --steps 20 --warmup 3 2>/dev/null | tail -1 | cut -c1-330 | tee -a $OUT/bench_train.txt
done
exit 0
