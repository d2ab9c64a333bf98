#!/bin/bash
# r03x: conv1_1's persistent grid size; launch timeline of the training step (fp32-accurate and bf16 modes)
set -u
TAG=${1:-r03x}
OUT=gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
T=disn_amd/csrc/libdisn_amd_tuning.so
for B in 16 1; do
DISN_AMD_LIB=$T KNOBS="conv11_wgs=512;conv11_wgs=1024;conv11_wgs=2048;conv11_wgs=4096;conv11_wgs=512" timeout 200 python tools/conv_stack_time.py $B 2>&1 | grep -v amdgpu.ids | tee -a $OUT/conv11_wgs.txt
done
for D in f32 bf16; do
  rm -rf /tmp/tr_$D
  (cd /tmp && timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/tr_$D -- python $GRAFT_REPO_ROOT/bench.py --workload train --train-dtype $D --steps 4 --warmup 2 > $GRAFT_REPO_ROOT/$OUT/train_$D.log 2>&1)
  F=$(find /tmp/tr_$D -name '*kernel_trace.csv' | head -1)
  python tools/trace_step.py $F resize_kernel $OUT/train_step_${D}_trace.txt
  tail -3 $OUT/train_step_${D}_trace.txt
done
exit 0
