mkdir -p gpurun_out/r4s
python -m pytest tests/test_kernels_gpu.py -x -q -k "chwn" > gpurun_out/r4s/pytest_chwn.txt 2>&1; tail -3 gpurun_out/r4s/pytest_chwn.txt
for G in 0 1; do echo "== LSPS_CHWN_GROUP=$G" >> gpurun_out/r4s/chwn.txt; LSPS_CHWN_GROUP=$G python tools/bench_chwn.py 128 768 2>&1 | grep -v amdgpu | cut -c1-95 >> gpurun_out/r4s/chwn.txt; done
cat gpurun_out/r4s/chwn.txt
for G in 0 1; do LSPS_CHWN_GROUP=$G GRAPHS=1 STEPS=100 python tools/bench_estimate.py 2>&1 | grep estimate; done
