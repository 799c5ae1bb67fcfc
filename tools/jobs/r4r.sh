mkdir -p gpurun_out/r4r
python -m pytest tests/test_parity_gpu.py -x -q -s -k "frozen or overlapped or hip_graph or two_trainers" > gpurun_out/r4r/pytest_sel.txt 2>&1; grep -E "arena address|passed|failed" gpurun_out/r4r/pytest_sel.txt
GRAPHS=1 STEPS=100 python tools/bench_estimate.py 2>&1 | grep estimate
