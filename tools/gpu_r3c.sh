#!/bin/bash
# round 3: where an eight-step call's time is now + submission sweep for short runs
set -u
TAG=${1:-r03c}
OUT=gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 600 python -m pytest tests/test_gpu_model.py -q --no-header -p no:cacheprovider -s -k "batch_invariant" > $OUT/pytest_model_quick.log 2>&1; echo "model quick exit $?"; grep "job \|passed\|failed" $OUT/pytest_model_quick.log | tail -8
(cd /tmp && timeout 200 rocprofv3 --kernel-trace --output-format csv -d /tmp/profb8_$TAG -o trace -- python $GRAFT_REPO_ROOT/bench.py --steps 24 --warmup 8 --no-extras --in-flight 1 --batch 8 --spinup-s 0 > /tmp/profb8_$TAG.log 2>&1; echo "rocprof b8 exit $?")
python tools/trace_step.py $(find /tmp/profb8_$TAG -name "*kernel_trace.csv") resize_kernel 2>/dev/null | grep -v "at::native\|rocclr_copy" > $OUT/infer_call_b8_trace.txt; cat $OUT/infer_call_b8_trace.txt
for v in "--batch 8 --in-flight 2" "--batch 8 --in-flight 3" "--batch 10 --in-flight 2" "--batch 5 --in-flight 4" "--batch 7 --in-flight 3" "--batch 20 --in-flight 1" "--batch 4 --in-flight 3" "--batch 6 --in-flight 2" "--batch 12 --in-flight 2"; do
  echo "variant --steps 20 --warmup 5 --balance 0 $v" | tee -a $OUT/bench_variants20.txt
  timeout 120 python bench.py --steps 20 --warmup 5 --balance 0 $v --no-extras 2>/dev/null | tail -1 | cut -c100-200 | tee -a $OUT/bench_variants20.txt
done
exit 0
