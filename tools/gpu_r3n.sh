#!/bin/bash
set -u
TAG=${1:-r03n}
OUT=gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
for rep in 1 2 3; do
for v in "--pack-short 0" "--pack-short 1" "--pack-short 1 --batch 16" "--pack-short 0 --batch 16" "--pack-short 1 --batch 12" "--pack-short 1 --batch 14" "--pack-short 1 --batch 15"; do
  echo "== --steps 20 --warmup 5 $v" | tee -a $OUT/bench_pack.txt
  BENCH_DEBUG=1 timeout 120 python bench.py --steps 20 --warmup 5 $v --no-extras 2>&1 | grep "enqueue\|main line" | tee -a $OUT/bench_pack.txt
done
done
exit 0
