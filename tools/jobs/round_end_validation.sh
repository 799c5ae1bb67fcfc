#!/bin/bash
# Round-end validation on ONE box, everything on the FINAL build:
#   1. the PMC passes of the dominant kernels (f32: wino4_f3x3 / wino4_w3x3; bf16: the C8 pair) -> gpurun_out/round_end/r6_traffic*.json,
#      stamped with the hash of the kernel sources (copy them to profiles/: bench.py refuses a file taken on other sources);
#   2. GPU suite;  3. the driver's bench command (reads the fresh traffic files);  4. config 5;  5. kernel stats of both.
# Counters only in the --pmc passes (never combined with other trace domains).
OUT=gpurun_out/round_end
mkdir -p $OUT
rm -rf gpurun_out/round_end_pmc_f32 gpurun_out/round_end_pmc_bf16
tools/pmc_pass.sh round_end_pmc_f32 tools/pmc_traffic.py "GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES" "FETCH_SIZE" "WRITE_SIZE"
tools/pmc_pass.sh round_end_pmc_bf16 tools/pmc_c8.py "GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES" "FETCH_SIZE" "WRITE_SIZE"
python tools/make_traffic_json.py f32 gpurun_out/round_end_pmc_f32/pmc_summary.txt > $OUT/r6_traffic.json
python tools/make_traffic_json.py bf16 gpurun_out/round_end_pmc_bf16/pmc_summary.txt > $OUT/r6_traffic_bf16.json
cp gpurun_out/round_end_pmc_f32/pmc_summary.txt $OUT/r6_pmc_wino4.txt
cp gpurun_out/round_end_pmc_bf16/pmc_summary.txt $OUT/r6_pmc_c8.txt
cp $OUT/r6_traffic.json $OUT/r6_traffic_bf16.json profiles/        # on the box: the bench below reads them; copy them home as well
python -m pytest tests -m gpu -x -q > $OUT/pytest_gpu.txt 2>&1; tail -4 $OUT/pytest_gpu.txt
python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench.json 2> $OUT/bench.err
python bench.py --exp nicvl --dtype bf16 --batch 256 --steps 10 --warmup 3 --no-extra --no-cpu-baseline > $OUT/config5.json 2> $OUT/config5.err
tools/prof_bench.sh round_end/prof_f32
tools/prof_bench.sh round_end/prof_c5 --exp nicvl --dtype bf16 --batch 256
python tools/show_bench.py $OUT/bench.json $OUT/config5.json | grep -E "value|extra|roof|cpu|traffic"
