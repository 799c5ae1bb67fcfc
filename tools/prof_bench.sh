#!/bin/bash
# Usage (on the GPU box, from the repo root): tools/prof_bench.sh <tag> [extra bench.py args]
# rocprofv3 --kernel-trace --stats over a short bench.py run; leaves gpurun_out/<tag>/{kernel_stats.txt,bench_under_rocprof.json}
REPO=$(cd "$(dirname "$0")/.." && pwd)
TAG=$1; shift
OUT=$REPO/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_$TAG
timeout 900 rocprofv3 --kernel-trace --stats -d /tmp/prof_$TAG -o run -- python $REPO/bench.py --steps 4 --warmup 1 --no-extra --no-cpu-baseline "$@" \
  > $OUT/bench_under_rocprof.json 2> $OUT/rocprof.err
db=$(find /tmp/prof_$TAG -name '*.db' | head -1)
if [ -n "$db" ]; then python $REPO/tools/rocprof_summary.py $db > $OUT/kernel_stats.txt; else echo "no db" > $OUT/kernel_stats.txt; find /tmp/prof_$TAG | head -20 >> $OUT/kernel_stats.txt; fi
find /tmp/prof_$TAG -name '*kernel_stats.csv' -exec cp {} $OUT/ \;
