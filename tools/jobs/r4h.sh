mkdir -p gpurun_out/r4h
python tools/est_two_graphs_probe.py > gpurun_out/r4h/two_graphs.txt 2>&1; grep -v amdgpu gpurun_out/r4h/two_graphs.txt | tail -12
python -m pytest tests/test_dist_gpu.py -x -q -k "hip_graphs" > gpurun_out/r4h/pytest_sel.txt 2>&1; tail -30 gpurun_out/r4h/pytest_sel.txt
