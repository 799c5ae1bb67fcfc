mkdir -p gpurun_out/r4n
python -m pytest tests -m gpu -x -q > gpurun_out/r4n/pytest_gpu.txt 2>&1; tail -4 gpurun_out/r4n/pytest_gpu.txt
python bench.py --steps 20 --warmup 5 > gpurun_out/r4n/bench.json 2> gpurun_out/r4n/bench.err; tail -c 300 gpurun_out/r4n/bench.json
LSPS_DEFER_GRADS=0 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r4n/bench_nodefer.json 2> gpurun_out/r4n/bench_nodefer.err
python bench.py --exp nicvl --dtype bf16 --batch 256 --steps 10 --warmup 3 --no-extra --no-cpu-baseline > gpurun_out/r4n/config5.json 2> gpurun_out/r4n/config5.err
LSPS_DEFER_GRADS=0 python bench.py --exp nicvl --dtype bf16 --batch 256 --steps 10 --warmup 3 --no-extra --no-cpu-baseline > gpurun_out/r4n/config5_nodefer.json 2> gpurun_out/r4n/config5_nodefer.err
python tools/show_bench.py gpurun_out/r4n/bench.json gpurun_out/r4n/bench_nodefer.json gpurun_out/r4n/config5.json gpurun_out/r4n/config5_nodefer.json 2>/dev/null | head -40
