# estimate3 step: kernel timeline (eager and replayed from the hipGraph) + kernel stats
mkdir -p gpurun_out/est_timeline
R=$PWD
cd /tmp && export TMPDIR=/tmp
for G in 0 1; do
  rm -rf /tmp/pe$G
  GRAPHS=$G STEPS=10 timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/pe$G -o run -- python $R/tools/bench_estimate.py > $R/gpurun_out/est_timeline/estimate_g$G.txt 2>&1
  f=$(find /tmp/pe$G -name '*kernel_trace.csv' | head -1)
  python $R/tools/timeline.py $f 2 --all > $R/gpurun_out/est_timeline/estimate3_timeline_g$G.txt 2>&1
done
cd $R
STEPS=50 python tools/bench_estimate.py > gpurun_out/est_timeline/estimate_plain.txt 2>&1
GRAPHS=1 STEPS=50 python tools/bench_estimate.py >> gpurun_out/est_timeline/estimate_plain.txt 2>&1
cat gpurun_out/est_timeline/estimate_plain.txt
