#!/bin/bash
# round-6 evidence run (same as round 5's script): GPU tests, bench (main line + extras, the driver's command), grid line, rocprof traces (call,
# step, conv stack), PMC traffic (FETCH / WRITE passes) and MFMA utilisation (its own pass) of the SAME build
set -u
TAG=${1:-r05x}; shift || true
WHAT=${*:-"smoke tests sweep bench ab grid prof pmc mfma"}
OUT=gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
has() { [[ " $WHAT " == *" $1 "* ]]; }
if has test; then
  timeout 300 python -m pytest tests/test_gpu_model.py -q -k "pipeline or standalone_layer_chain or cfg2" --no-header -p no:cacheprovider > $OUT/pytest_quick.log 2>&1; echo "quick tests exit $?"; tail -3 $OUT/pytest_quick.log
fi
if has smoke; then
  timeout 600 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; echo "smoke exit $?"; tail -1 $OUT/smoke.log
fi
if has tests; then
  timeout 1500 python -m pytest tests -m gpu -q -rA --no-header -p no:cacheprovider > $OUT/pytest_gpu.log 2>&1; echo "pytest exit $?" | tee -a $OUT/pytest_gpu.log; grep -v "^PASSED" $OUT/pytest_gpu.log | tail -15
fi
if has sweep; then   # the FULL trained-like sweep (48 weight sets): tests/test_gpu_sweep.py writes gpurun_out/sweep_full.json
  DISN_SWEEP=full timeout 1500 python -m pytest tests/test_gpu_sweep.py -m gpu -q -rA -s --no-header -p no:cacheprovider > $OUT/pytest_sweep_full.log 2>&1; echo "sweep exit $?" | tee -a $OUT/pytest_sweep_full.log
  cp gpurun_out/sweep_full.json $OUT/sweep_full.json 2>/dev/null; grep "parity sweep" $OUT/pytest_sweep_full.log | tail -4
fi
if has bench; then
  timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_driver_cmd.json 2> $OUT/bench_driver_cmd.err; echo "bench (driver's command) exit $?"; tail -2 $OUT/bench_driver_cmd.err
  timeout 600 python bench.py > $OUT/bench.json 2> $OUT/bench.err; echo "bench exit $?"; tail -3 $OUT/bench.err
  for v in "--batch 1 --in-flight 1" "--batch 1 --in-flight 3" "--batch 2 --in-flight 2" "--batch 4 --in-flight 1" "--batch 4 --in-flight 2" "--batch 8 --in-flight 1" "--batch 8 --in-flight 2" "--batch 16 --in-flight 1" "--steps 20 --warmup 5" "--strict" "--strict --steps 20 --warmup 5"; do
    echo "variant $v" | tee -a $OUT/bench_variants.txt
    timeout 120 python bench.py $(case "$v" in *--steps*) ;; *) echo --steps 240 --warmup 24;; esac) $v --no-extras 2>/dev/null | tail -1 | cut -c1-260 | tee -a $OUT/bench_variants.txt
  done
fi
if has ab; then
  # A/B on this box with the tuning build: round 3's layer-by-layer point MLP of a batched call (fused_small=0) against
  # the fused small-set kernels (default), steady state and the driver's command; and the driver's 20 steps cut evenly
  for k in "fused_small=1" "fused_small=0"; do
    echo "knobs $k" | tee -a $OUT/bench_ab.txt
    DISN_AMD_LIB=disn_amd/csrc/libdisn_amd_tuning.so KNOBS=$k timeout 150 python tools/bench_knobs.py --steps 256 --warmup 32 --no-extras 2>/dev/null | tail -1 | cut -c1-200 | tee -a $OUT/bench_ab.txt
    DISN_AMD_LIB=disn_amd/csrc/libdisn_amd_tuning.so KNOBS=$k timeout 150 python tools/bench_knobs.py --steps 20 --warmup 5 --no-extras 2>/dev/null | tail -1 | cut -c1-200 | tee -a $OUT/bench_ab.txt
  done
  for bt in 4 8; do for k in "fused_small=1" "fused_small=0"; do
    echo "batch $bt x 2 in flight, knobs $k" | tee -a $OUT/bench_ab.txt
    DISN_AMD_LIB=disn_amd/csrc/libdisn_amd_tuning.so KNOBS=$k timeout 150 python tools/bench_knobs.py --steps 256 --warmup 32 --no-extras --batch $bt 2>/dev/null | tail -1 | cut -c1-200 | tee -a $OUT/bench_ab.txt
  done; done
  echo "balance 1" | tee -a $OUT/bench_ab.txt
  timeout 150 python bench.py --steps 20 --warmup 5 --no-extras --balance 1 2>/dev/null | tail -1 | cut -c1-200 | tee -a $OUT/bench_ab.txt
fi
if has grid; then
  timeout 300 python bench.py --workload grid --steps 3 --warmup 1 > $OUT/bench_grid1.json 2> $OUT/bench_grid1.err; echo "grid exit $?"; tail -c 400 $OUT/bench_grid1.json
fi
if has prof; then
  # (a) the main line's own command (default --batch / --in-flight), kernel stats
  (cd /tmp && timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/profm_$TAG -o trace -- python $GRAFT_REPO_ROOT/bench.py --steps 40 --warmup 8 --no-extras > /tmp/profm_$TAG.log 2>&1; echo "rocprof main exit $?")
  for f in $(find /tmp/profm_$TAG -name "*kernel_stats.csv"); do cp $f $OUT/bench_kernel_stats.csv; done
  # (a2) one call of eight steps at a time: the launch sequence of a batched call
  (cd /tmp && timeout 200 rocprofv3 --kernel-trace --output-format csv -d /tmp/profb4_$TAG -o trace -- python $GRAFT_REPO_ROOT/bench.py --steps 32 --warmup 16 --no-extras --in-flight 1 --batch 16 --spinup-s 0 > /tmp/profb4_$TAG.log 2>&1; echo "rocprof b4 exit $?")
  python tools/trace_step.py $(find /tmp/profb4_$TAG -name "*kernel_trace.csv") resize_kernel - 2 2>/dev/null | grep -v "at::native\|rocclr_copy" > $OUT/infer_call_b16_trace.txt; tail -2 $OUT/infer_call_b16_trace.txt
  # (b) one step at a time: the launch sequence of a step
  (cd /tmp && timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$TAG -o trace -- python $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 3 --no-extras --in-flight 1 --batch 1 > /tmp/prof_$TAG.log 2>&1; echo "rocprof exit $?")
  for f in $(find /tmp/prof_$TAG -name "*kernel_stats.csv"); do cp $f $OUT/bench_single_kernel_stats.csv; done
  python tools/trace_step.py $(find /tmp/prof_$TAG -name "*kernel_trace.csv") resize_kernel 2>/dev/null | grep -v "at::native\|rocclr_copy" > $OUT/infer_step_trace.txt; tail -3 $OUT/infer_step_trace.txt
  (cd /tmp && timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/profc_$TAG -o trace -- python $GRAFT_REPO_ROOT/tools/conv_stack_time.py > /tmp/profc_$TAG.log 2>&1; echo "rocprof conv stack exit $?")
  for f in $(find /tmp/profc_$TAG -name "*kernel_stats.csv"); do cp $f $OUT/conv_stack_kernel_stats.csv; done
  python tools/trace_step.py $(find /tmp/profc_$TAG -name "*kernel_trace.csv") resize_kernel 2>/dev/null > $OUT/conv_stack_trace.txt; tail -2 $OUT/conv_stack_trace.txt
  (cd /tmp && timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/profc4_$TAG -o trace -- python $GRAFT_REPO_ROOT/tools/conv_stack_time.py 16 > /tmp/profc4_$TAG.log 2>&1; echo "rocprof conv stack x8 exit $?")
  for f in $(find /tmp/profc4_$TAG -name "*kernel_stats.csv"); do cp $f $OUT/conv_stack_b16_kernel_stats.csv; done
  python tools/trace_step.py $(find /tmp/profc4_$TAG -name "*kernel_trace.csv") resize_kernel 2>/dev/null > $OUT/conv_stack_b16_trace.txt; tail -2 $OUT/conv_stack_b16_trace.txt
  (cd /tmp && timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/profk_$TAG -o trace -- python $GRAFT_REPO_ROOT/tools/prof_kernels.py > /tmp/profk_$TAG.log 2>&1; echo "rocprof kernels exit $?")
  for f in $(find /tmp/profk_$TAG -name "*kernel_stats.csv"); do cp $f $OUT/kernels_kernel_stats.csv; done
fi
if has pmc; then
  cd /tmp
  for C in FETCH_SIZE WRITE_SIZE; do
    PROF_STEPS=1 PMC_BUILD=$TAG timeout 150 rocprofv3 --pmc $C --kernel-trace --output-format csv -d /tmp/pmc_${TAG}_$C -o pmc -- python $GRAFT_REPO_ROOT/tools/prof_kernels.py > $GRAFT_REPO_ROOT/$OUT/pmc_$C.log 2>&1
    echo "pmc $C exit $?"
    for f in $(find /tmp/pmc_${TAG}_$C -name "*counter_collection.csv"); do cp $f $GRAFT_REPO_ROOT/$OUT/pmc_$C.csv; done
  done
  cd $GRAFT_REPO_ROOT
  PMC_BUILD=$TAG python tools/pmc_traffic.py $OUT > $OUT/pmc_traffic.json && head -14 $OUT/pmc_traffic.json
fi
if has mfma; then
  cd /tmp
  timeout 200 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d /tmp/pmc_${TAG}_mfma -o pmc -- python $GRAFT_REPO_ROOT/tools/prof_mfma.py > $GRAFT_REPO_ROOT/$OUT/pmc_mfma.log 2>&1
  echo "pmc mfma exit $?"
  for f in $(find /tmp/pmc_${TAG}_mfma -name "*counter_collection.csv"); do cp $f $GRAFT_REPO_ROOT/$OUT/pmc_MFMA_BUSY.csv; done
  cd $GRAFT_REPO_ROOT
  python tools/pmc_mfma.py $OUT/pmc_MFMA_BUSY.csv > $OUT/pmc_mfma.txt && cat $OUT/pmc_mfma.txt
fi
exit 0
