#!/bin/bash
# Usage: tools/pmc_pass.sh <outdir under gpurun_out> <workload.py> "CTR1 CTR2 ..." ["CTR ..." ...]
# One rocprofv3 --pmc pass per counter group (never combined with other trace domains), summarised per kernel.
REPO=$(cd "$(dirname "$0")/.." && pwd)
OUT=$REPO/gpurun_out/$1; WL=$2; shift 2
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
i=0
for grp in "$@"; do
  i=$((i+1))
  rm -rf /tmp/pmc_$i
  timeout 600 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d /tmp/pmc_$i -- python $REPO/$WL > $OUT/pass$i.log 2>&1
  f=$(find /tmp/pmc_$i -name '*counter_collection.csv' | head -1)
  echo "## pass $i: $grp" >> $OUT/pmc_summary.txt
  if [ -n "$f" ]; then python $REPO/tools/rocprof_summary.py $f | grep -A14 -E "wino|igemm|axpy|chwn|c1_|c8|x3" >> $OUT/pmc_summary.txt; else echo "no csv (see pass$i.log)" >> $OUT/pmc_summary.txt; tail -5 $OUT/pass$i.log >> $OUT/pmc_summary.txt; fi
done
