#!/bin/bash
# round 3, first GPU call: parity of the batched convolution (conv_h2w.hip), its per-layer times at eight images per
# call against the round-2 kernels, its phase timeline
set -u
TAG=${1:-r03a}
OUT=gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
T=disn_amd/csrc/libdisn_amd_tuning.so
timeout 900 python -m pytest tests/test_gpu_conv_h2.py -q -rA --no-header -p no:cacheprovider -x > $OUT/pytest_conv_h2.log 2>&1; echo "conv tests exit $?"; grep -v "^PASSED" $OUT/pytest_conv_h2.log | tail -25
(cd /tmp && timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/profc8_$TAG -o trace -- python $GRAFT_REPO_ROOT/tools/conv_stack_time.py 8 > /tmp/profc8_$TAG.log 2>&1; echo "rocprof conv stack x8 exit $?"; tail -3 /tmp/profc8_$TAG.log)
for f in $(find /tmp/profc8_$TAG -name "*kernel_stats.csv"); do cp $f $OUT/conv_stack_b8_kernel_stats.csv; done
python tools/trace_step.py $(find /tmp/profc8_$TAG -name "*kernel_trace.csv") resize_kernel 2>/dev/null > $OUT/conv_stack_b8_trace.txt; cat $OUT/conv_stack_b8_trace.txt
(cd /tmp && timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/profc4_$TAG -o trace -- python $GRAFT_REPO_ROOT/tools/conv_stack_time.py 4 > /tmp/profc4_$TAG.log 2>&1; echo "rocprof conv stack x4 exit $?")
python tools/trace_step.py $(find /tmp/profc4_$TAG -name "*kernel_trace.csv") resize_kernel 2>/dev/null > $OUT/conv_stack_b4_trace.txt; tail -16 $OUT/conv_stack_b4_trace.txt
DISN_AMD_LIB=$T KNOBS="conv_wide_min=1073741824;conv_wide_min=4" timeout 200 python tools/conv_stack_time.py 8 > $OUT/conv_stack_b8_wide_vs_r02.txt 2>&1; cat $OUT/conv_stack_b8_wide_vs_r02.txt
DISN_AMD_LIB=$T STAMP_CK=16 timeout 200 python tools/conv_h2_stamps.py 8 > $OUT/conv_h2w_stamps_b8.txt 2>&1; cat $OUT/conv_h2w_stamps_b8.txt
timeout 600 python -m pytest tests/test_gpu_model.py -q --no-header -p no:cacheprovider -k "batched or pipeline or cfg2 or standalone" > $OUT/pytest_model_quick.log 2>&1; echo "model quick exit $?"; tail -15 $OUT/pytest_model_quick.log
exit 0
