mkdir -p gpurun_out/r4e
R=$PWD
for ORDER in reg_first feat_first; do
  echo "== split, order $ORDER" >> gpurun_out/r4e/est.txt
  LSPS_EST_ORDER=$ORDER STEPS=100 python tools/bench_estimate.py 2>&1 | grep estimate >> gpurun_out/r4e/est.txt
  LSPS_EST_ORDER=$ORDER GRAPHS=1 STEPS=100 python tools/bench_estimate.py 2>&1 | grep estimate >> gpurun_out/r4e/est.txt
  (cd /tmp; export TMPDIR=/tmp; rm -rf /tmp/pe; LSPS_EST_ORDER=$ORDER GRAPHS=1 STEPS=10 timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/pe -o run -- python $R/tools/bench_estimate.py > /dev/null 2>&1
   python $R/tools/timeline.py $(find /tmp/pe -name '*kernel_trace.csv' | head -1) 2 --all > $R/gpurun_out/r4e/timeline_${ORDER}_g1.txt 2>&1)
  python tools/timeline_windows.py gpurun_out/r4e/timeline_${ORDER}_g1.txt 500 > gpurun_out/r4e/windows_${ORDER}.txt
done
echo "== no split wino4 (LSPS_WINO4_SPLIT=0)" >> gpurun_out/r4e/est.txt
LSPS_WINO4_SPLIT=0 GRAPHS=1 STEPS=100 python tools/bench_estimate.py 2>&1 | grep estimate >> gpurun_out/r4e/est.txt
cat gpurun_out/r4e/est.txt
python tools/gradient_criterion.py --impl hip > gpurun_out/r4e/gradient_criterion_hip.txt 2>&1
LSPS_WINO=2 python tools/gradient_criterion.py --impl hip --sets full,extra > gpurun_out/r4e/gradient_criterion_hip_always.txt 2>&1
LSPS_WINO=0 python tools/gradient_criterion.py --impl hip --sets full,extra > gpurun_out/r4e/gradient_criterion_hip_off.txt 2>&1
