# kernel trace (csv) of a short default bench run: every launch of the bs=128 pretrain step with its grid and duration
mkdir -p gpurun_out/pretrain_trace
R=$PWD
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/ptr
timeout 900 rocprofv3 --kernel-trace --output-format csv -d /tmp/ptr -o run -- python $R/bench.py --steps 2 --warmup 1 --no-extra --no-cpu-baseline ${BENCH_ARGS} > $R/gpurun_out/pretrain_trace/bench.json 2> $R/gpurun_out/pretrain_trace/err.txt
f=$(find /tmp/ptr -name '*kernel_trace.csv' | head -1)
cp $f $R/gpurun_out/pretrain_trace/kernel_trace.csv
ls -la $R/gpurun_out/pretrain_trace/
