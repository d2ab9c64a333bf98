#!/bin/bash
# r03ac: training tests + bench after the weight-gradient changes (interleave rule, f16 split)
set -u
TAG=${1:-r03ac}
OUT=gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 900 python -m pytest tests/test_gpu_train.py tests/test_gpu_bf16.py -q --no-header -p no:cacheprovider -x 2>&1 | tail -4 | tee $OUT/pytest_train.log
for D in f32 bf16; do
  timeout 300 python bench.py --workload train --train-dtype $D 2>&1 | grep -v amdgpu.ids | tail -1 > $OUT/bench_train_$D.json
  python -c "
import json
d=json.load(open('$OUT/bench_train_$D.json'))
print('$D', d['value'], d['unit'], d['ms_per_step'], 'ms/step')"
done
rm -rf /tmp/tr_f32
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/tr_f32 -- python $GRAFT_REPO_ROOT/bench.py --workload train --train-dtype f32 --steps 4 --warmup 2 > $GRAFT_REPO_ROOT/$OUT/train_f32.log 2>&1)
F=$(find /tmp/tr_f32 -name '*kernel_trace.csv' | head -1)
python tools/trace_step.py $F pack_multi $OUT/train_step_f32_trace.txt
tail -1 $OUT/train_step_f32_trace.txt
exit 0
