#!/bin/bash
# round 3: schedules of a batched call's local fold2/conv1 (0 / 2 / 3 K ranges) with the 16-loads-in-flight gather
set -u
TAG=${1:-r03g}
OUT=gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
T=disn_amd/csrc/libdisn_amd_tuning.so
timeout 900 python -m pytest tests/test_gpu_model.py -q --no-header -p no:cacheprovider -s -k "encode_query or batch_invariant or cfg2" > $OUT/pytest_model.log 2>&1; echo "model tests exit $?"; grep "job \|passed\|failed\|^E " $OUT/pytest_model.log | tail -8
for K in 0 2 3; do
  for v in "--steps 20 --warmup 5" "--steps 240 --warmup 24" "--steps 240 --warmup 24 --in-flight 1"; do
    echo "l4_ranges=$K $v" | tee -a $OUT/bench_l4.txt
    DISN_AMD_LIB=$T KNOBS=l4_ranges=$K timeout 120 python tools/bench_knobs.py $v --balance 0 --no-extras 2>/dev/null | tail -1 | cut -c100-200 | tee -a $OUT/bench_l4.txt
  done
done
(cd /tmp && timeout 200 rocprofv3 --kernel-trace --output-format csv -d /tmp/profb8_$TAG -o trace -- python $GRAFT_REPO_ROOT/bench.py --steps 24 --warmup 8 --no-extras --in-flight 1 --batch 8 --spinup-s 0 > /tmp/profb8_$TAG.log 2>&1; echo "rocprof b8 exit $?")
python tools/trace_step.py $(find /tmp/profb8_$TAG -name "*kernel_trace.csv") resize_kernel 2>/dev/null | grep -v "at::native\|rocclr_copy" > $OUT/infer_call_b8_trace.txt; cat $OUT/infer_call_b8_trace.txt
exit 0
