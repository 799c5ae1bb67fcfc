mkdir -p gpurun_out/r4f
R=$PWD
for ORDER in chain feat_first; do for FZ in 0 1; do
  echo "== order $ORDER frozen_packs_off=$FZ" >> gpurun_out/r4f/est.txt
  LSPS_NO_FROZEN_PACKS=$FZ LSPS_EST_ORDER=$ORDER GRAPHS=1 STEPS=100 python tools/bench_estimate.py 2>&1 | grep estimate >> gpurun_out/r4f/est.txt
done; done
LSPS_EST_ORDER=chain STEPS=100 python tools/bench_estimate.py 2>&1 | grep estimate >> gpurun_out/r4f/est.txt
cat gpurun_out/r4f/est.txt
(cd /tmp; export TMPDIR=/tmp; rm -rf /tmp/pe; LSPS_EST_ORDER=chain GRAPHS=1 STEPS=10 timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/pe -o run -- python $R/tools/bench_estimate.py > /dev/null 2>&1
 python $R/tools/timeline.py $(find /tmp/pe -name '*kernel_trace.csv' | head -1) 2 --all > $R/gpurun_out/r4f/timeline_chain_g1.txt 2>&1)
python tools/timeline_windows.py gpurun_out/r4f/timeline_chain_g1.txt 500 > gpurun_out/r4f/windows_chain.txt; cat gpurun_out/r4f/windows_chain.txt
python -m pytest tests -m gpu -x -q > gpurun_out/r4f/pytest_gpu.txt 2>&1; tail -6 gpurun_out/r4f/pytest_gpu.txt
