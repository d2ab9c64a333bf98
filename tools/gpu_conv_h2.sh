#!/bin/bash
# conv_h2 bring-up round: parity tests, in-kernel phase stamps (tuning build), per-layer kernel durations from a trace
TAG=${1:-r02c}; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_conv_h2.py -x -q --no-header -p no:cacheprovider > $OUT/pytest_conv_h2.log 2>&1; echo "pytest exit $?"; tail -5 $OUT/pytest_conv_h2.log
DISN_AMD_LIB=disn_amd/csrc/libdisn_amd_tuning.so timeout 300 python tools/conv_h2_stamps.py > $OUT/conv_h2_stamps.txt 2>&1; grep -E "cin|chunk1|last|total|setup|prologue|epilogue|reduce" $OUT/conv_h2_stamps.txt
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/p1 -o t -- python $GRAFT_REPO_ROOT/tools/conv_h2_time.py > $GRAFT_REPO_ROOT/$OUT/conv_h2_time.txt 2>&1)
python tools/trace_summary.py $(find /tmp/p1 -name "*kernel_trace.csv") conv_h2_kernel | tee $OUT/conv_h2_trace_summary.txt
python tools/trace_summary.py $(find /tmp/p1 -name "*kernel_trace.csv") conv1_1 | tee -a $OUT/conv_h2_trace_summary.txt
timeout 300 python bench.py --steps 50 --warmup 5 --no-extras 2>&1 | tail -2 | cut -c1-300
