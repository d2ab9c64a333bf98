#!/bin/bash
# One GPU-box round: fused-kernel tests first (fail fast), whole GPU suite, smoke, bench lines, rocprof
# kernel trace, PMC traffic passes.  Outputs -> gpurun_out/<tag>/    usage: tools/gpu_round2.sh [tag] [what...]
set -u
TAG=${1:-r02a}; shift || true
WHAT=${*:-"fused tests smoke bench grid prof pmc"}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
rocminfo 2>/dev/null | grep -E "Marketing Name|Compute Unit|Max Clock" | head -6 > $OUT/device.txt
nproc >> $OUT/device.txt; grep -m1 "model name" /proc/cpuinfo >> $OUT/device.txt
has() { [[ " $WHAT " == *" $1 "* ]]; }
if has fused; then
  echo "== fused tests"; timeout 900 python -m pytest tests/test_gpu_fused.py -x -q -rA --no-header -p no:cacheprovider > $OUT/pytest_fused.log 2>&1
  echo "fused exit $?" | tee -a $OUT/pytest_fused.log; tail -25 $OUT/pytest_fused.log
fi
if has tests; then
  echo "== pytest -m gpu"; timeout 1800 python -m pytest tests -m gpu -q -rA --no-header -p no:cacheprovider > $OUT/pytest_gpu.log 2>&1
  echo "pytest exit $?" | tee -a $OUT/pytest_gpu.log; tail -40 $OUT/pytest_gpu.log
fi
if has smoke; then
  echo "== smoke"; timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; echo "smoke exit $?" | tee -a $OUT/smoke.log; tail -3 $OUT/smoke.log
fi
if has bench; then
  echo "== bench"; timeout 900 python bench.py --steps 50 --warmup 5 > $OUT/bench.json 2> $OUT/bench.err; echo "bench exit $?"; tail -c 2500 $OUT/bench.json; tail -5 $OUT/bench.err
fi
if has grid; then
  echo "== bench grid N=1"; timeout 600 python bench.py --workload grid --steps 3 --warmup 1 > $OUT/bench_grid1.json 2> $OUT/bench_grid1.err; echo "exit $?"; tail -c 1500 $OUT/bench_grid1.json; tail -3 $OUT/bench_grid1.err
  echo "== bench grid N=1 unfused"; timeout 600 python bench.py --workload grid --unfused --steps 2 --warmup 1 > $OUT/bench_grid1_unfused.json 2> $OUT/bench_grid1_unfused.err; echo "exit $?"; tail -c 600 $OUT/bench_grid1_unfused.json
  echo "== bench grid --gpus 2 (ranks share the GPU: plumbing)"; timeout 900 python bench.py --gpus 2 --workload grid --grid-images 2 --steps 2 --warmup 1 > $OUT/bench_grid2.json 2> $OUT/bench_grid2.err; echo "exit $?"; tail -c 1500 $OUT/bench_grid2.json; tail -3 $OUT/bench_grid2.err
  echo "== nccl with two ranks on one GPU?"; timeout 300 python bench.py --gpus 2 --workload grid --grid-res 32 --grid-images 2 --steps 1 --warmup 1 --dist-backend nccl > $OUT/bench_grid2_nccl.json 2> $OUT/bench_grid2_nccl.err; echo "exit $?"; tail -c 300 $OUT/bench_grid2_nccl.json; grep -i "duplicate\|error" $OUT/bench_grid2_nccl.err | head -3
fi
if has prof; then
  echo "== rocprof"; (cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$TAG -o trace -- python $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 3 --no-extras > /tmp/prof_$TAG.log 2>&1; echo "rocprof exit $?")
  for f in $(find /tmp/prof_$TAG -name "*kernel_stats.csv"); do cp $f $OUT/bench_kernel_stats.csv; done
  for f in $(find /tmp/prof_$TAG -name "*kernel_trace.csv"); do cp $f $OUT/bench_kernel_trace.csv; done
  head -16 $OUT/bench_kernel_stats.csv 2>/dev/null | cut -c1-150
  (cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/profk_$TAG -o trace -- python $GRAFT_REPO_ROOT/tools/prof_kernels.py > /tmp/profk_$TAG.log 2>&1; echo "rocprof kernels exit $?")
  for f in $(find /tmp/profk_$TAG -name "*kernel_stats.csv"); do cp $f $OUT/kernels_kernel_stats.csv; done
  for f in $(find /tmp/profk_$TAG -name "*kernel_trace.csv"); do cp $f $OUT/kernels_kernel_trace.csv; done
  grep -i "fused\|gather" $OUT/kernels_kernel_stats.csv | cut -c1-160
fi
if has pmc; then
  echo "== pmc"; bash tools/gpu_pmc_traffic.sh $TAG
fi
exit 0
