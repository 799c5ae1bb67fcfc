mkdir -p gpurun_out/r4m
TORCH_FR_BUFFER_SIZE=2000 TORCH_NCCL_TRACE_BUFFER_SIZE=2000 python tools/dp_drain_probe.py 2>&1 | grep -v "amdgpu\|^\[\|RCCL\|HIP ver\|ROCm\|Hostname\|Librccl" > gpurun_out/r4m/drain_probe.txt; cat gpurun_out/r4m/drain_probe.txt
python -m pytest tests/test_dist_gpu.py -x -q > gpurun_out/r4m/pytest_dist.txt 2>&1; tail -3 gpurun_out/r4m/pytest_dist.txt
LSPS_FORCE_DP=1 python bench.py --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/r4m/bench_1rank_rccl.json 2> gpurun_out/r4m/bench_1rank_rccl.err; tail -c 1500 gpurun_out/r4m/bench_1rank_rccl.json; tail -3 gpurun_out/r4m/bench_1rank_rccl.err | cut -c1-300
bash tools/jobs/r4l_pmc.sh
