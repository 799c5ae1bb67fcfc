mkdir -p gpurun_out/r4i
python tools/stream_overlap_probe.py 2>&1 | grep -v amdgpu > gpurun_out/r4i/overlap_default.txt
GPU_MAX_HW_QUEUES=8 python tools/stream_overlap_probe.py 2>&1 | grep -v amdgpu > gpurun_out/r4i/overlap_q8.txt
GPU_MAX_HW_QUEUES=2 python tools/stream_overlap_probe.py 2>&1 | grep -v amdgpu > gpurun_out/r4i/overlap_q2.txt
cat gpurun_out/r4i/overlap_default.txt gpurun_out/r4i/overlap_q8.txt gpurun_out/r4i/overlap_q2.txt
