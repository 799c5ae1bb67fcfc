# PMC passes for the traffic files of the final build (counters only: never combined with other trace domains)
rm -rf gpurun_out/r4l_f32 gpurun_out/r4l_bf16
tools/pmc_pass.sh r4l_f32 tools/pmc_traffic.py "GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES" "FETCH_SIZE" "WRITE_SIZE"
tools/pmc_pass.sh r4l_bf16 tools/pmc_c8.py "GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES" "FETCH_SIZE" "WRITE_SIZE"
python tools/make_traffic_json.py f32 gpurun_out/r4l_f32/pmc_summary.txt > gpurun_out/r4l_f32/r4_traffic.json
python tools/make_traffic_json.py bf16 gpurun_out/r4l_bf16/pmc_summary.txt > gpurun_out/r4l_bf16/r4_traffic_bf16.json
head -30 gpurun_out/r4l_f32/r4_traffic.json
