mkdir -p gpurun_out/r4b
python tools/dp_drain_probe.py > gpurun_out/r4b/drain_probe_default.txt 2>&1
TORCH_NCCL_TRACE_BUFFER_SIZE=2000 python tools/dp_drain_probe.py > gpurun_out/r4b/drain_probe_2000.txt 2>&1
cat gpurun_out/r4b/drain_probe_default.txt gpurun_out/r4b/drain_probe_2000.txt
bash tools/jobs/r4b_estimate_timeline.sh
