mkdir -p gpurun_out/r4k
python -m pytest tests -m gpu -x -q > gpurun_out/r4k/pytest_gpu.txt 2>&1; tail -6 gpurun_out/r4k/pytest_gpu.txt
python tools/gradient_criterion.py --impl hip > gpurun_out/r4k/gradient_criterion_hip.txt 2>&1
GRAPHS=1 STEPS=100 python tools/bench_estimate.py 2>&1 | grep estimate
python bench.py --gpus 2 --backend gloo --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/r4k/bench_2ranks_gloo_one_gpu.json 2> gpurun_out/r4k/bench_2ranks.err; tail -c 800 gpurun_out/r4k/bench_2ranks_gloo_one_gpu.json
LSPS_FORCE_DP=1 python bench.py --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/r4k/bench_1rank_rccl.json 2> gpurun_out/r4k/bench_1rank_rccl.err; tail -c 1200 gpurun_out/r4k/bench_1rank_rccl.json
