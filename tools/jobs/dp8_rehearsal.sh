#!/bin/bash
# BASELINE config 4's WORLD SIZE on a 1-GPU box: eight ranks over gloo sharing the one GPU (RCCL refuses two
# ranks on one device).  Proves the launcher, shard_batch, the out-of-band capture agreement, the learned bucket schedule and the
# scalar all-reduce at world = 8; says nothing about xGMI.  16 samples per domain and rank (8 x 128 would not fit one GPU's time).
mkdir -p gpurun_out/dp8
python bench.py --gpus 8 --backend gloo --batch 16 --steps 6 --warmup 3 --no-cpu-baseline > gpurun_out/dp8/bench_8ranks_gloo_one_gpu.json 2> gpurun_out/dp8/bench.err
tail -c 1500 gpurun_out/dp8/bench_8ranks_gloo_one_gpu.json; tail -5 gpurun_out/dp8/bench.err
