mkdir -p gpurun_out/r4p
python -m pytest tests -m gpu -x -q > gpurun_out/r4p/pytest_gpu.txt 2>&1; tail -4 gpurun_out/r4p/pytest_gpu.txt
GRAPHS=1 STEPS=100 python tools/bench_estimate.py 2>&1 | grep estimate
LSPS_WINO4_SPLIT=0 GRAPHS=1 STEPS=100 python tools/bench_estimate.py 2>&1 | grep estimate
LSPS_WINO4_SPLIT=0 LSPS_NO_FROZEN_PACKS=1 LSPS_EST_SPLIT_BACKWARD=0 GRAPHS=1 STEPS=100 python tools/bench_estimate.py 2>&1 | grep estimate
STEPS=100 python tools/bench_estimate.py 2>&1 | grep estimate
python bench.py --steps 20 --warmup 5 > gpurun_out/r4p/bench.json 2> gpurun_out/r4p/bench.err
R=$PWD; (cd /tmp; export TMPDIR=/tmp; rm -rf /tmp/pe; GRAPHS=1 STEPS=20 timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/pe -o run -- python $R/tools/bench_estimate.py > $R/gpurun_out/r4p/estimate_under_rocprof.txt 2>&1
 db=$(find /tmp/pe -name '*.db' | head -1); python $R/tools/rocprof_summary.py $db > $R/gpurun_out/r4p/estimate3_kernel_stats.txt)
python tools/show_bench.py gpurun_out/r4p/bench.json | head -10
head -8 gpurun_out/r4p/estimate3_kernel_stats.txt
