#!/bin/bash
# the three-limb forward kernel with a 3-deep image ring and counted waits (LSPS_X3_RING=1) against the
# two-stage double buffer, same box: correctness (tests/test_x3_gpu.py under the ring) and per-layer timings (tools/check_x3s2.py).
mkdir -p gpurun_out/r6_x3_ring
LSPS_X3_RING=1 python -m pytest tests/test_x3_gpu.py -x -q > gpurun_out/r6_x3_ring/tests_ring.txt 2>&1
tail -3 gpurun_out/r6_x3_ring/tests_ring.txt
for r in 0 1 0 1; do
  LSPS_X3_RING=$r python tools/check_x3s2.py > gpurun_out/r6_x3_ring/check_ring$r.txt 2>&1
  echo "== LSPS_X3_RING=$r"; grep -E "ms|err" gpurun_out/r6_x3_ring/check_ring$r.txt | head -40
done
