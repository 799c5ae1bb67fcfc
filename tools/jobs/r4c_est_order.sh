mkdir -p gpurun_out/r4c
python -m pytest tests/test_dist_gpu.py -x -q -k "hip_graphs or resume" > gpurun_out/r4c/pytest_dist.txt 2>&1; tail -3 gpurun_out/r4c/pytest_dist.txt
for ORDER in side_first main_first; do for PRIO in 0 -1; do
  echo "== order $ORDER prio $PRIO" >> gpurun_out/r4c/est.txt
  LSPS_EST_ORDER=$ORDER LSPS_SIDE_PRIO=$PRIO STEPS=100 python tools/bench_estimate.py 2>&1 | grep estimate >> gpurun_out/r4c/est.txt
  LSPS_EST_ORDER=$ORDER LSPS_SIDE_PRIO=$PRIO GRAPHS=1 STEPS=100 python tools/bench_estimate.py 2>&1 | grep estimate >> gpurun_out/r4c/est.txt
done; done
cat gpurun_out/r4c/est.txt
R=$PWD; cd /tmp; export TMPDIR=/tmp
rm -rf /tmp/pe; LSPS_EST_ORDER=main_first GRAPHS=1 STEPS=10 timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/pe -o run -- python $R/tools/bench_estimate.py > /dev/null 2>&1
python $R/tools/timeline.py $(find /tmp/pe -name '*kernel_trace.csv' | head -1) 2 --all > $R/gpurun_out/r4c/timeline_main_first_g1.txt 2>&1
