#!/bin/bash
set -u
TAG=${1:-r03u}
OUT=gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
T=disn_amd/csrc/libdisn_amd_tuning.so
DISN_AMD_LIB=$T STAMP_CK=16 timeout 200 python tools/conv_h2_stamps.py 16 2>&1 | grep -v amdgpu.ids > $OUT/conv_h2w_stamps_b16.txt; cat $OUT/conv_h2w_stamps_b16.txt
timeout 600 python -m pytest tests/test_gpu_dense_h2.py -q --no-header -p no:cacheprovider 2>&1 | tail -2
exit 0
