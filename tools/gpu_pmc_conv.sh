#!/bin/bash
# SQ counters of single conv_h2 layers -> gpurun_out/<tag>/pmc_conv_*.csv
TAG=${1:-pmc}; OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
rocprofv3 -L 2>/dev/null | grep -oE "\b(SQ|TA|TCP|TD|TCC)_[A-Z0-9_]+" | sort -u > $OUT/counters_avail.txt; wc -l $OUT/counters_avail.txt
i=0
for SET in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES" "SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_VALU SQ_INSTS_VMEM SQ_INST_CYCLES_VMEM" "TA_BUSY_avr TA_TA_BUSY_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum"; do
  i=$((i+1))
  for L in "512 512 28" "512 512 14" "256 256 56"; do
    n=$(echo $L | tr ' ' '_')
    timeout 300 rocprofv3 --pmc $SET --kernel-trace --output-format csv -d /tmp/pc_${i}_$n -o p -- python $GRAFT_REPO_ROOT/tools/conv_h2_one.py $L > $OUT/pmc_conv_${i}_$n.log 2>&1
    f=$(find /tmp/pc_${i}_$n -name "*counter_collection.csv" | head -1)
    [ -n "$f" ] && grep conv_h2_kernel $f | head -40 > $OUT/pmc_conv_${i}_$n.csv && head -1 $f > $OUT/pmc_conv_header.csv
  done
done
ls $OUT
