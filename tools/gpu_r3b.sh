#!/bin/bash
# round 3: conv_h2w iteration -- parity, per-layer times at 8 images, stamps, the step's model tests, bench variants
set -u
TAG=${1:-r03b}; shift || true
WHAT=${*:-"conv stamps model bench"}
OUT=gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
T=disn_amd/csrc/libdisn_amd_tuning.so
has() { [[ " $WHAT " == *" $1 "* ]]; }
if has conv; then
  timeout 900 python -m pytest tests/test_gpu_conv_h2.py -q --no-header -p no:cacheprovider > $OUT/pytest_conv_h2.log 2>&1; echo "conv tests exit $?"; tail -4 $OUT/pytest_conv_h2.log
  (cd /tmp && timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/profc8_$TAG -o trace -- python $GRAFT_REPO_ROOT/tools/conv_stack_time.py 8 > /tmp/profc8_$TAG.log 2>&1; echo "rocprof conv stack x8 exit $?"; grep "conv stack" /tmp/profc8_$TAG.log)
  for f in $(find /tmp/profc8_$TAG -name "*kernel_stats.csv"); do cp $f $OUT/conv_stack_b8_kernel_stats.csv; done
  python tools/trace_step.py $(find /tmp/profc8_$TAG -name "*kernel_trace.csv") resize_kernel 2>/dev/null > $OUT/conv_stack_b8_trace.txt; cat $OUT/conv_stack_b8_trace.txt
fi
if has stamps; then
  DISN_AMD_LIB=$T STAMP_CK=16 timeout 200 python tools/conv_h2_stamps.py 8 2>&1 | grep -v amdgpu.ids > $OUT/conv_h2w_stamps_b8.txt; cat $OUT/conv_h2w_stamps_b8.txt
fi
if has model; then
  timeout 900 python -m pytest tests/test_gpu_model.py -q --no-header -p no:cacheprovider -s -k "batched or pipeline or cfg2 or standalone" > $OUT/pytest_model_quick.log 2>&1; echo "model quick exit $?"; grep -v "^$" $OUT/pytest_model_quick.log | tail -12
fi
if has bench; then
  for v in "--steps 20 --warmup 5 --balance 0" "--steps 20 --warmup 5 --balance 1" "--steps 240 --warmup 24" "--steps 240 --warmup 24 --batch 8 --in-flight 1" "--steps 240 --warmup 24 --batch 16 --in-flight 2"; do
    echo "variant $v" | tee -a $OUT/bench_variants.txt
    timeout 120 python bench.py $v --no-extras 2>/dev/null | tail -1 | cut -c1-200 | tee -a $OUT/bench_variants.txt
  done
fi
if has tests; then
  timeout 2400 python -m pytest tests -m gpu -q -rA --no-header -p no:cacheprovider > $OUT/pytest_gpu.log 2>&1; echo "pytest exit $?" | tee -a $OUT/pytest_gpu.log; grep -v "^PASSED" $OUT/pytest_gpu.log | tail -25
fi
exit 0
