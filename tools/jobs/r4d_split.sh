mkdir -p gpurun_out/r4d
for SP in 0 1; do
  echo "== split backward $SP" >> gpurun_out/r4d/est.txt
  LSPS_EST_SPLIT_BACKWARD=$SP STEPS=100 python tools/bench_estimate.py 2>&1 | grep estimate >> gpurun_out/r4d/est.txt
  LSPS_EST_SPLIT_BACKWARD=$SP GRAPHS=1 STEPS=100 python tools/bench_estimate.py 2>&1 | grep estimate >> gpurun_out/r4d/est.txt
  LSPS_EST_SPLIT_BACKWARD=$SP MODE=4 GRAPHS=1 STEPS=100 python tools/bench_estimate.py 2>&1 | grep estimate >> gpurun_out/r4d/est.txt
done
cat gpurun_out/r4d/est.txt
python -m pytest tests/test_parity_gpu.py tests/test_dist_gpu.py tests/test_driver_gpu.py -x -q > gpurun_out/r4d/pytest.txt 2>&1; tail -5 gpurun_out/r4d/pytest.txt
R=$PWD; cd /tmp; export TMPDIR=/tmp
rm -rf /tmp/pe; GRAPHS=1 STEPS=10 timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/pe -o run -- python $R/tools/bench_estimate.py > /dev/null 2>&1
python $R/tools/timeline.py $(find /tmp/pe -name '*kernel_trace.csv' | head -1) 2 --all > $R/gpurun_out/r4d/timeline_split_g1.txt 2>&1
