mkdir -p gpurun_out/r4t
python -m pytest tests -m gpu -x -q > gpurun_out/r4t/pytest_gpu.txt 2>&1; tail -4 gpurun_out/r4t/pytest_gpu.txt
python bench.py --steps 20 --warmup 5 > gpurun_out/r4t/bench.json 2> gpurun_out/r4t/bench.err
LSPS_CHWN_GROUP=0 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extra > gpurun_out/r4t/bench_nogroup.json 2> gpurun_out/r4t/bench_nogroup.err
tools/prof_bench.sh r4t/prof_f32
python tools/show_bench.py gpurun_out/r4t/bench.json gpurun_out/r4t/bench_nogroup.json | grep -E "value|extra|roof|chwn"
