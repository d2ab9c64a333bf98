#!/bin/bash
set -u
TAG=${1:-r03m}
OUT=gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 900 python -m pytest tests/test_gpu_conv_h2.py tests/test_gpu_dense_h2.py -q --no-header -p no:cacheprovider > $OUT/pytest_units.log 2>&1; echo "unit tests exit $?"; tail -3 $OUT/pytest_units.log
timeout 300 python tools/dense_h2w_time.py 2>&1 | grep -v amdgpu.ids | tee $OUT/dense_h2w_time.txt
timeout 900 python -m pytest tests/test_gpu_model.py -q --no-header -p no:cacheprovider -s > $OUT/pytest_model.log 2>&1; echo "model tests exit $?"; grep "job \|passed\|failed\|^E " $OUT/pytest_model.log | tail -8
(cd /tmp && timeout 200 rocprofv3 --kernel-trace --output-format csv -d /tmp/profc8_$TAG -o trace -- python $GRAFT_REPO_ROOT/tools/conv_stack_time.py 8 > /tmp/profc8_$TAG.log 2>&1; grep "conv stack" /tmp/profc8_$TAG.log)
python tools/trace_step.py $(find /tmp/profc8_$TAG -name "*kernel_trace.csv") resize_kernel 2>/dev/null | head -4
(cd /tmp && timeout 200 rocprofv3 --kernel-trace --output-format csv -d /tmp/profb8_$TAG -o trace -- python $GRAFT_REPO_ROOT/bench.py --steps 24 --warmup 8 --no-extras --in-flight 1 --batch 8 --spinup-s 0 > /tmp/profb8_$TAG.log 2>&1; echo "rocprof b8 exit $?")
python tools/trace_step.py $(find /tmp/profb8_$TAG -name "*kernel_trace.csv") resize_kernel 2>/dev/null | grep -v "at::native\|rocclr_copy" > $OUT/infer_call_b8_trace.txt; cat $OUT/infer_call_b8_trace.txt
for v in "--steps 20 --warmup 5" "--steps 20 --warmup 5" "--steps 240 --warmup 24" "--steps 240 --warmup 24 --in-flight 1"; do
  echo "variant $v" | tee -a $OUT/bench_variants.txt
  timeout 120 python bench.py $v --balance 0 --no-extras 2>&1 | grep "main line" | tee -a $OUT/bench_variants.txt
done
exit 0
