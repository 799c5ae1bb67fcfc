# A/B of the X3 launch plan (LSPS_X3_PLAN=0: the round-5 rule; 1: cost model + linear walk) on one box
mkdir -p gpurun_out/x3plan
O=gpurun_out/x3plan
timeout 900 python -m pytest tests/test_x3_gpu.py -x -q -m gpu > $O/pytest_x3.txt 2>&1; tail -3 $O/pytest_x3.txt
for P in 0 1 0 1; do
  echo "== LSPS_X3_PLAN=$P" >> $O/estimate.txt
  LSPS_X3_PLAN=$P GRAPHS=1 STEPS=100 python tools/bench_estimate.py 2>/dev/null >> $O/estimate.txt
  LSPS_X3_PLAN=$P STEPS=100 python tools/bench_estimate.py 2>/dev/null >> $O/estimate.txt
done
cat $O/estimate.txt
for P in 0 1; do
  LSPS_X3_PLAN=$P python bench.py --steps 8 --warmup 3 --no-extra --no-cpu-baseline 2>/dev/null > $O/bench_plan$P.json
  python - <<PY
import json
d=json.loads(open('$O/bench_plan$P.json').read().strip().splitlines()[-1])
print('plan $P', d['ms_per_step'], d['roofline']['three_limb_stride2_family']['ms_per_step'])
PY
done
