cd /tmp; export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/kt_q -o kt -- python $GRAFT_REPO_ROOT/tools/query_trace.py > /dev/null 2>&1
f=$(find /tmp/kt_q -name "*kernel_trace.csv" | head -1)
python $GRAFT_REPO_ROOT/tools/trace_step.py $f pt_embed_kernel $GRAFT_REPO_ROOT/gpurun_out/query65536_trace.txt
cat $GRAFT_REPO_ROOT/gpurun_out/query65536_trace.txt
