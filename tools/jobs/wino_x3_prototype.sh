#!/bin/bash
# the three-limb F(4x4,3x3) prototype's numbers.  (1) the register-level probe (bf16 MFMA beside the transform /
# split VALU mix, K = 8 vs K = 16 forms), (2) the main-loop skeleton of the one design that fits registers and LDS, timed as a whole
# and per phase, (3) its PMC split (separate --pmc passes, counters only).  Output: gpurun_out/r6_wino_x3/.
REPO=$(cd "$(dirname "$0")/../.." && pwd)
OUT=$REPO/gpurun_out/r6_wino_x3
mkdir -p $OUT
$REPO/tools/probes/ovl16 > $OUT/probe.txt 2>&1
$REPO/tools/probes/wino_x3_skeleton > $OUT/skeleton.txt 2>&1
cd /tmp && export TMPDIR=/tmp
i=0
for grp in "GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS" \
           "SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_INSTS_VMEM_RD SQ_WAIT_INST_LDS" "FETCH_SIZE"; do
  i=$((i+1))
  rm -rf /tmp/pmcw_$i
  timeout 300 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d /tmp/pmcw_$i -- $REPO/tools/probes/wino_x3_skeleton > $OUT/pass$i.log 2>&1
  f=$(find /tmp/pmcw_$i -name '*counter_collection.csv' | head -1)
  echo "## pass $i: $grp" >> $OUT/pmc.txt
  if [ -n "$f" ]; then python $REPO/tools/rocprof_summary.py $f | grep -A8 "skeleton" >> $OUT/pmc.txt; else echo "no csv" >> $OUT/pmc.txt; tail -3 $OUT/pass$i.log >> $OUT/pmc.txt; fi
done
cat $OUT/skeleton.txt
head -60 $OUT/pmc.txt
