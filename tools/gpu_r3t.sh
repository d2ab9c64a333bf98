#!/bin/bash
set -u
TAG=${1:-r03t}
OUT=gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
T=disn_amd/csrc/libdisn_amd_tuning.so
for M in 16384 32768; do
DISN_AMD_LIB=$T KNOBS="densew_m64=0,densew_c128=1;densew_m64=1,densew_c128=1;densew_m64=0,densew_c128=0;densew_m64=1,densew_c128=0" timeout 300 python tools/dense_h2w_time.py $M 2>&1 | grep -v amdgpu.ids | tee -a $OUT/dense_h2w_knobs.txt
done
exit 0
