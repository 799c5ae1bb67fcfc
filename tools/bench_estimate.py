#!/usr/bin/env python
"""Times the literal estimate3 step (LSPSTrainer.post_update(mode=3), reference lsps_trainer.py:220-262) at
bs=128 per domain; used under rocprofv3 to see where its time goes."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import bench  # noqa: E402

hp = bench.load_hp()
import lsps_amd.trainers as trainers  # noqa: E402
tr = trainers.LSPSTrainer(hp)
tr.cuda(0)
dev = torch.device('cuda', 0)
b = bench.make_device_batch(int(os.environ.get('BS', '128')), dev)
mode = int(os.environ.get('MODE', '3'))
step = lambda: tr.post_update(b['xa'], b['la'], b['xb'], b['lb'], b['ca'], b['cb'], mode, hp)  # noqa: E731
if os.environ.get('GRAPHS') == '1':
    tr.use_graphs(True)
for _ in range(3):
    step()
torch.cuda.synchronize()
t0 = time.perf_counter()
K = int(os.environ.get('STEPS', '20'))
for _ in range(K):
    step()
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / K
print("estimate%d bs=%d: %.3f ms/step = %.1f steps/s" % (mode, b['xa'].shape[0], 1e3 * dt, 1.0 / dt))
