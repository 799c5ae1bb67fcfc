mkdir -p gpurun_out/r4a
python -m pytest tests -m gpu -x -q > gpurun_out/r4a/pytest_gpu.txt 2>&1; echo "pytest rc $?" >> gpurun_out/r4a/pytest_gpu.txt
tail -5 gpurun_out/r4a/pytest_gpu.txt
python tools/gradient_criterion.py --impl hip > gpurun_out/r4a/gradient_criterion_hip.txt 2>&1
python bench.py --steps 20 --warmup 5 > gpurun_out/r4a/bench.json 2> gpurun_out/r4a/bench.err
tail -c 600 gpurun_out/r4a/bench.json
python bench.py --exp nicvl --dtype bf16 --batch 256 --steps 10 --warmup 3 --no-extra --no-cpu-baseline > gpurun_out/r4a/config5.json 2> gpurun_out/r4a/config5.err
tools/prof_bench.sh r4a/prof_c5 --exp nicvl --dtype bf16 --batch 256
python bench.py --steps 3 --warmup 1 --no-extra --cpu-baseline-bs128 > gpurun_out/r4a/bench_cpu128.json 2> gpurun_out/r4a/bench_cpu128.err
cat gpurun_out/cpu_baseline_bs128.json
