#!/bin/bash
# two PMC passes (one counter each, kernel-trace only) over the bounded workload -> gpurun_out/<tag>/
set -u
TAG=${1:-pmc}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp
for C in FETCH_SIZE WRITE_SIZE; do
  PROF_STEPS=1 timeout 600 rocprofv3 --pmc $C --kernel-trace --output-format csv -d /tmp/pmc_${TAG}_$C -o pmc -- python $GRAFT_REPO_ROOT/tools/prof_kernels.py > $OUT/pmc_$C.log 2>&1
  echo "pmc $C exit $?"
  for f in $(find /tmp/pmc_${TAG}_$C -name "*counter_collection.csv"); do cp $f $OUT/pmc_$C.csv; done
done
python $GRAFT_REPO_ROOT/tools/pmc_traffic.py $OUT > $OUT/pmc_traffic.json && head -12 $OUT/pmc_traffic.json
