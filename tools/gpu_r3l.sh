#!/bin/bash
set -u
TAG=${1:-r03l}
OUT=gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
for v in "--batch 8 --in-flight 2:3" "--batch 8 --in-flight 1:3" "--batch 10 --in-flight 1:2"; do
  a=${v%%:*}; n=${v##*:}
  (cd /tmp && rm -rf /tmp/p20 && timeout 200 rocprofv3 --kernel-trace --output-format csv -d /tmp/p20 -o trace -- python $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 5 --balance 0 $a --no-extras > /tmp/p20.log 2>&1; grep "main line" /tmp/p20.log)
  echo "== $a" | tee -a $OUT/region20.txt
  python tools/trace_region.py $(find /tmp/p20 -name "*kernel_trace.csv") resize_kernel $n | tee -a $OUT/region20.txt
done
exit 0
