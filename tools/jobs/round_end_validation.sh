# round-end validation on one box: GPU suite, the driver's bench command, the same bench with the round-4 dispatch of the stride-2 layers, kernel stats
mkdir -p gpurun_out/round_end
python -m pytest tests -m gpu -x -q > gpurun_out/round_end/pytest_gpu.txt 2>&1; tail -4 gpurun_out/round_end/pytest_gpu.txt
python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/round_end/bench.json 2> gpurun_out/round_end/bench.err
LSPS_X3=0 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extra > gpurun_out/round_end/bench_nox3.json 2> gpurun_out/round_end/bench_nox3.err
tools/prof_bench.sh round_end/prof_f32
python tools/show_bench.py gpurun_out/round_end/bench.json gpurun_out/round_end/bench_nox3.json | grep -E "value|extra|roof|cpu"
