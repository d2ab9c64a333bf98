#!/bin/bash
set -u
TAG=${1:-r03h}
OUT=gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
T=disn_amd/csrc/libdisn_amd_tuning.so
DISN_AMD_LIB=$T KNOBS="gather_l16=0;gather_l16=1" timeout 120 python tools/gather_time.py 8 2>&1 | grep -v amdgpu.ids | tee $OUT/gather_time.txt
DISN_AMD_LIB=$T KNOBS="gather_l16=0;gather_l16=1" timeout 120 python tools/gather_time.py 1 2>&1 | grep -v amdgpu.ids | tee -a $OUT/gather_time.txt
for K in "l4_ranges=0" "l4_ranges=2" "l4_ranges=3" "l4_ranges=0,gather_l16=0" "l4_ranges=2,gather_l16=0" "l4_ranges=3,gather_l16=0"; do
  for v in "--steps 20 --warmup 5" "--steps 240 --warmup 24" "--steps 240 --warmup 24 --in-flight 1"; do
    echo "$K $v" | tee -a $OUT/bench_l4.txt
    DISN_AMD_LIB=$T KNOBS=$K timeout 120 python tools/bench_knobs.py $v --balance 0 --no-extras 2>$OUT/err.txt | tail -1 | cut -c100-200 | tee -a $OUT/bench_l4.txt
    tail -2 $OUT/err.txt | grep -i "error\|Traceback" 
  done
done
tail -5 $OUT/err.txt
exit 0
