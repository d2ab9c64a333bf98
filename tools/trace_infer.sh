cd /tmp; export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/kt_b -o kt -- python $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 3 --no-extras > /dev/null 2>&1
f=$(find /tmp/kt_b -name "*kernel_trace.csv" | head -1)
python $GRAFT_REPO_ROOT/tools/trace_step.py $f resize_kernel\<1\> $GRAFT_REPO_ROOT/gpurun_out/infer_step_trace.txt
tail -1 $GRAFT_REPO_ROOT/gpurun_out/infer_step_trace.txt
