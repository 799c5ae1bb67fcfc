mkdir -p gpurun_out/r4o
python -m pytest tests -m gpu -x -q > gpurun_out/r4o/pytest_gpu.txt 2>&1; tail -4 gpurun_out/r4o/pytest_gpu.txt
python bench.py --steps 20 --warmup 5 > gpurun_out/r4o/bench.json 2> gpurun_out/r4o/bench.err; tail -c 300 gpurun_out/r4o/bench.json
tools/prof_bench.sh r4o/prof_f32
tools/prof_bench.sh r4o/prof_c5 --exp nicvl --dtype bf16 --batch 256
R=$PWD; (cd /tmp; export TMPDIR=/tmp; rm -rf /tmp/pe; GRAPHS=1 STEPS=20 timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/pe -o run -- python $R/tools/bench_estimate.py > $R/gpurun_out/r4o/estimate_under_rocprof.txt 2>&1
 db=$(find /tmp/pe -name '*.db' | head -1); python $R/tools/rocprof_summary.py $db > $R/gpurun_out/r4o/estimate3_kernel_stats.txt)
GRAPHS=1 STEPS=100 python tools/bench_estimate.py 2>&1 | grep estimate
STEPS=100 python tools/bench_estimate.py 2>&1 | grep estimate
python tools/show_bench.py gpurun_out/r4o/bench.json | head -12
