#!/bin/bash
# A/B of bench.py's --in-flight on the driver's command (20 steps, 5 warm-up), interleaved runs: tools/driver_cmd_ab.sh [runs=8] "<args A>" "<args B>"
export TMPDIR=/tmp
R=${1:-8}; A=${2:---in-flight 2}; B=${3:---in-flight 3}
for r in $(seq 1 $R); do
  for v in "$A" "$B"; do
    x=$(timeout 100 python bench.py --steps 20 --warmup 5 --no-extras $v 2>/dev/null | tail -1 | python -c 'import sys,json; print(round(json.loads(sys.stdin.read())["value"]/1e6,3))')
    echo "$v : $x"
  done
done | sort | awk -F' : ' '{a[$1]=a[$1]" "$2; s[$1]+=$2; n[$1]++} END {for (k in a) print k, "|", a[k], "| mean", s[k]/n[k]}'
