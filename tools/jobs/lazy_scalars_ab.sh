#!/bin/bash
# Round 6: loss scalars published asynchronously (LSPS_LAZY_SCALARS, default on) against the synchronous copy per update method, same box.
mkdir -p gpurun_out/r6_lazy
python -m pytest tests/test_driver_gpu.py tests/test_parity_gpu.py -x -q -k "scalars or hip_graph or n32" > gpurun_out/r6_lazy/tests.txt 2>&1; tail -3 gpurun_out/r6_lazy/tests.txt
for v in 1 0 1 0; do
  LSPS_LAZY_SCALARS=$v python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r6_lazy/bench_lazy$v.json 2> gpurun_out/r6_lazy/bench_lazy$v.err
  python - <<PY
import json
j=[json.loads(l) for l in open('gpurun_out/r6_lazy/bench_lazy$v.json') if l.startswith('{')][-1]
e=j['other_workloads']
print('LSPS_LAZY_SCALARS=$v  pretrain %.2f ms/step   estimate3 %s   bs32 %s' % (j['ms_per_step'], {k:(v.get('ms_per_step'), v.get('eager_ms_per_step')) for k,v in e.items() if k.startswith('estimate3')}, {k:v.get('ms_per_step') for k,v in e.items() if 'bs32' in k}))
PY
done
