#!/bin/bash
set -u
TAG=${1:-r03p}
OUT=gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
for rep in 1 2; do
for v in "" "--batch 20" "--batch 20 --in-flight 1" "--batch 10" "--batch 16 --in-flight 3"; do
  echo "== --steps 20 --warmup 5 $v" | tee -a $OUT/bench20.txt
  timeout 120 python bench.py --steps 20 --warmup 5 $v --no-extras 2>&1 | grep "main line" | tee -a $OUT/bench20.txt
done
done
for v in "" "--batch 32" "--batch 24"; do
  echo "== --steps 320 --warmup 32 $v" | tee -a $OUT/bench20.txt
  timeout 120 python bench.py --steps 320 --warmup 32 $v --no-extras 2>&1 | grep "main line" | tee -a $OUT/bench20.txt
done
(cd /tmp && timeout 200 rocprofv3 --kernel-trace --output-format csv -d /tmp/profc16_$TAG -o trace -- python $GRAFT_REPO_ROOT/tools/conv_stack_time.py 16 > /tmp/profc16_$TAG.log 2>&1; grep "conv stack" /tmp/profc16_$TAG.log)
python tools/trace_step.py $(find /tmp/profc16_$TAG -name "*kernel_trace.csv") resize_kernel 2>/dev/null | tee $OUT/conv_stack_b16_trace.txt
timeout 900 python -m pytest tests/test_gpu_model.py -q --no-header -p no:cacheprovider -s -k "threshold or batch_invariant" > $OUT/pytest_sel.log 2>&1; echo "tests exit $?"; grep "vs float64\|passed\|failed\|^E " $OUT/pytest_sel.log | tail -8
exit 0
