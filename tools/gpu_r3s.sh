#!/bin/bash
set -u
TAG=${1:-r03s}
OUT=gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 900 python -m pytest tests/test_gpu_conv_h2.py -q --no-header -p no:cacheprovider -k "conv1_1 or standalone" > $OUT/pytest_c11.log 2>&1; echo "conv1_1 tests exit $?"; tail -3 $OUT/pytest_c11.log
(cd /tmp && timeout 200 rocprofv3 --kernel-trace --output-format csv -d /tmp/profc16_$TAG -o trace -- python $GRAFT_REPO_ROOT/tools/conv_stack_time.py 16 > /tmp/profc16_$TAG.log 2>&1; grep "conv stack" /tmp/profc16_$TAG.log)
python tools/trace_step.py $(find /tmp/profc16_$TAG -name "*kernel_trace.csv") resize_kernel 2>/dev/null | tee $OUT/conv_stack_b16_trace.txt
exit 0
