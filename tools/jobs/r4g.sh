mkdir -p gpurun_out/r4g
python tools/est_two_graphs_probe.py > gpurun_out/r4g/two_graphs.txt 2>&1; grep -v amdgpu gpurun_out/r4g/two_graphs.txt | tail -12
R=$PWD
(cd /tmp; export TMPDIR=/tmp; rm -rf /tmp/pe; timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/pe -o run -- python $R/tools/est_two_graphs_probe.py > /dev/null 2>&1
 python $R/tools/timeline.py $(find /tmp/pe -name '*kernel_trace.csv' | head -1) 330 --all > $R/gpurun_out/r4g/timeline_two.txt 2>&1)
python tools/timeline_windows.py gpurun_out/r4g/timeline_two.txt 500 > gpurun_out/r4g/windows_two.txt; cat gpurun_out/r4g/windows_two.txt
python -m pytest tests/test_dist_gpu.py tests/test_parity_gpu.py tests/test_kernels_gpu.py -x -q -k "hip_graphs or frozen or reduction_split or overlapped or resume" > gpurun_out/r4g/pytest_sel.txt 2>&1; tail -6 gpurun_out/r4g/pytest_sel.txt
