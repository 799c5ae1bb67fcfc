# literal estimate3 step (post_update mode 3, bs=128): per-kernel stats of the graph replay and of the eager step, + the plain timings
mkdir -p gpurun_out/est_stats
R=$PWD
cd /tmp && export TMPDIR=/tmp
for G in 1 0; do
  rm -rf /tmp/pes$G
  GRAPHS=$G STEPS=30 timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/pes$G -o run -- python $R/tools/bench_estimate.py > $R/gpurun_out/est_stats/estimate_g$G.txt 2>&1
  db=$(find /tmp/pes$G -name '*.db' | head -1)
  if [ -n "$db" ]; then python $R/tools/rocprof_summary.py $db > $R/gpurun_out/est_stats/estimate3_kernel_stats_g$G.txt; fi
done
cd $R
STEPS=100 python tools/bench_estimate.py > gpurun_out/est_stats/estimate_plain.txt 2>&1
GRAPHS=1 STEPS=100 python tools/bench_estimate.py >> gpurun_out/est_stats/estimate_plain.txt 2>&1
cat gpurun_out/est_stats/estimate_plain.txt
head -30 gpurun_out/est_stats/estimate3_kernel_stats_g1.txt | cut -c1-150
