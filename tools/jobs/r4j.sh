mkdir -p gpurun_out/r4j
python -m pytest tests -m gpu -x -q > gpurun_out/r4j/pytest_gpu.txt 2>&1; tail -6 gpurun_out/r4j/pytest_gpu.txt
python bench.py --steps 20 --warmup 5 > gpurun_out/r4j/bench.json 2> gpurun_out/r4j/bench.err; tail -c 1500 gpurun_out/r4j/bench.json
