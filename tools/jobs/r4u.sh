mkdir -p gpurun_out/r4u
for F in 2 3 4; do echo "== LSPS_CHWN_FILL=$F" >> gpurun_out/r4u/chwn.txt; LSPS_CHWN_FILL=$F python tools/bench_chwn.py 128 2>&1 | grep -v amdgpu | cut -c1-95 >> gpurun_out/r4u/chwn.txt; done
cat gpurun_out/r4u/chwn.txt
for F in 2 4; do LSPS_CHWN_FILL=$F GRAPHS=1 STEPS=100 python tools/bench_estimate.py 2>&1 | grep estimate; done
