mkdir -p gpurun_out/r4q
LSPS_CHWN_MIN_N=1 python tools/bench_chwn.py 128 144 16 256 2>&1 | grep -v amdgpu > gpurun_out/r4q/chwn.txt; cat gpurun_out/r4q/chwn.txt
