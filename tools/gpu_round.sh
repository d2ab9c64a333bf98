#!/bin/bash
# One GPU-box round: parity tests, smoke, bench, rocprof kernel trace.  Outputs -> gpurun_out/
# usage: tools/gpu_round.sh [tag]
set -u
TAG=${1:-r01}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
rocminfo 2>/dev/null | grep -E "Marketing Name|Compute Unit|Max Clock" | head -6 > $OUT/device.txt
nproc >> $OUT/device.txt; grep -m1 "model name" /proc/cpuinfo >> $OUT/device.txt
echo "== pytest -m gpu" ; timeout 1500 python -m pytest tests -m gpu -q -rA --no-header -p no:cacheprovider > $OUT/pytest_gpu.log 2>&1
echo "pytest exit $?" | tee -a $OUT/pytest_gpu.log
tail -40 $OUT/pytest_gpu.log
echo "== smoke" ; timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; echo "smoke exit $?" | tee -a $OUT/smoke.log; tail -3 $OUT/smoke.log
echo "== bench" ; timeout 900 python bench.py --steps 50 --warmup 5 > $OUT/bench.json 2> $OUT/bench.err; echo "bench exit $?"; tail -c 3000 $OUT/bench.json; tail -5 $OUT/bench.err
echo "== rocprof" ; (cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$TAG -o trace -- python $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 3 --no-extras > /tmp/prof_$TAG.log 2>&1; echo "rocprof exit $?")
for f in $(find /tmp/prof_$TAG -name "*kernel_stats.csv"); do cp $f $OUT/bench_kernel_stats.csv; done
for f in $(find /tmp/prof_$TAG -name "*kernel_trace.csv"); do cp $f $OUT/bench_kernel_trace.csv; done
head -16 $OUT/bench_kernel_stats.csv 2>/dev/null | cut -c1-150
exit 0
