#!/bin/bash
# five PMC passes (one counter set each, kernel-trace only) + one kernel-trace/stats pass over
# tools/prof_kernels.py -> gpurun_out/<tag>/ ; summarise with tools/pmc_summary.py + tools/pmc_traffic.py
set -u
TAG=${1:-pmc}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp
PROF_STEPS=1 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt2_$TAG -o kt -- python $GRAFT_REPO_ROOT/tools/prof_kernels.py > $OUT/kt2.log 2>&1
for f in $(find /tmp/kt2_$TAG -name "*kernel_stats.csv"); do cp $f $OUT/kernels_kernel_stats.csv; done
for C in FETCH_SIZE WRITE_SIZE "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"; do
  N=$(echo $C | tr ' ' '_')
  PROF_STEPS=1 timeout 300 rocprofv3 --pmc $C --kernel-trace --output-format csv -d /tmp/pmc_${TAG}_$N -o pmc -- python $GRAFT_REPO_ROOT/tools/prof_kernels.py > $OUT/pmc_$N.log 2>&1
  echo "pmc $N exit $?"
  for f in $(find /tmp/pmc_${TAG}_$N -name "*counter_collection.csv"); do cp $f $OUT/pmc_$N.csv; done
done
python $GRAFT_REPO_ROOT/tools/pmc_traffic.py $OUT > $OUT/pmc_traffic.json
python $GRAFT_REPO_ROOT/tools/pmc_summary.py $OUT > $OUT/pmc_summary.txt
wc -l $OUT/pmc_summary.txt
