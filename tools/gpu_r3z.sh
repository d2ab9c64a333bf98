#!/bin/bash
# r03z: launch timeline of the training step with the conv_h2 data gradients
set -u
TAG=${1:-r03z}
OUT=gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 600 python -m pytest tests/test_gpu_train.py -q --no-header -p no:cacheprovider -x -k "data_gradient or conv3x3" 2>&1 | tail -3
for D in f32 bf16; do
  rm -rf /tmp/tr_$D
  (cd /tmp && timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/tr_$D -- python $GRAFT_REPO_ROOT/bench.py --workload train --train-dtype $D --steps 4 --warmup 2 > $GRAFT_REPO_ROOT/$OUT/train_$D.log 2>&1)
  F=$(find /tmp/tr_$D -name '*kernel_trace.csv' | head -1)
  python tools/trace_step.py $F pack_multi $OUT/train_step_${D}_trace.txt
  tail -2 $OUT/train_step_${D}_trace.txt
  timeout 300 python bench.py --workload train --train-dtype $D 2>&1 | grep -v amdgpu.ids | tail -1 > $OUT/bench_train_$D.json
  python -c "
import json
d=json.load(open('$OUT/bench_train_$D.json'))
print('$D', d['value'], d['unit'], d['ms_per_step'], 'ms/step')"
done
exit 0
