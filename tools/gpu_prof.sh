#!/bin/bash
# rocprofv3 passes: kernel trace + stats of bench.py, then PMC passes (one counter set per run,
# never combined with tracing domains other than kernel-trace) on the bounded workload.
set -u
TAG=${1:-r01}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt_$TAG -o kt -- python $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 3 --no-extras > $OUT/kt.log 2>&1
echo "kt exit $?"; find /tmp/kt_$TAG -type f | head -20
for f in $(find /tmp/kt_$TAG -name "*kernel_stats.csv"); do cp $f $OUT/bench_kernel_stats.csv; done
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt2_$TAG -o kt -- python $GRAFT_REPO_ROOT/tools/prof_kernels.py > $OUT/kt2.log 2>&1
for f in $(find /tmp/kt2_$TAG -name "*kernel_stats.csv"); do cp $f $OUT/kernels_kernel_stats.csv; done
for f in $(find /tmp/kt2_$TAG -name "*kernel_trace.csv"); do cp $f $OUT/kernels_kernel_trace.csv; done
for C in FETCH_SIZE WRITE_SIZE "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"; do
  N=$(echo $C | tr ' ' '_')
  PROF_STEPS=1 timeout 600 rocprofv3 --pmc $C --kernel-trace --output-format csv -d /tmp/pmc_${TAG}_$N -o pmc -- python $GRAFT_REPO_ROOT/tools/prof_kernels.py > $OUT/pmc_$N.log 2>&1
  echo "pmc $N exit $?"
  for f in $(find /tmp/pmc_${TAG}_$N -name "*counter_collection.csv"); do cp $f $OUT/pmc_$N.csv; done
done
ls -la $OUT
head -25 $OUT/bench_kernel_stats.csv
