#!/usr/bin/env python
"""Per-iteration deviation of the HIP trainer's loss scalars from the CPU oracle's over a training trajectory
(tests/golden/cases.py: run_trajectory; tests/test_trajectory_gpu.py holds the bound).  Usage: python tools/trajectory_report.py [tiny|full] [f32|bf16]"""
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [REPO, os.path.join(REPO, 'tests'), os.path.join(REPO, 'tests', 'golden')]
import numpy as np   # noqa: E402
import torch         # noqa: E402
import cases         # noqa: E402
from oracle import lsps_ref   # noqa: E402
import lsps_amd.trainers as prod   # noqa: E402

config = sys.argv[1] if len(sys.argv) > 1 else 'tiny'
mode = sys.argv[2] if len(sys.argv) > 2 else 'f32'          # 'bf16': the HIP trainer in the bf16 math mode (BASELINE config 5's arithmetic)
kw = dict() if config == 'tiny' else dict(n=4, n_pre=5, n_est=5, held_out=16, cadence=2)
torch.set_num_threads(8)
O = cases.NativeAdapter(lsps_ref, 'cpu', trainer_kwargs=dict(literal=False))
A = cases.NativeAdapter(prod, 'cuda')
ref = cases.run_trajectory(O, config, lsps_ref, **kw)
pert = [cases.run_trajectory(O, config, lsps_ref, perturb=p, **kw) for p in (1e-5, 3e-5)]
env = cases.trajectory_envelope(ref, pert)
from lsps_amd import ops   # noqa: E402
ops.set_math_mode(mode)
try:
    got = cases.run_trajectory(A, config, lsps_ref, **kw)
finally:
    ops.set_math_mode('f32')
print('HIP trainer math mode:', mode)
names = sorted(set(n for it in ref['scalars'].values() for n in it))
print('%-16s' % 'iteration' + ''.join('%15s' % n[-14:] for n in names))
for it, k in enumerate(ref['scalars']):
    row = '%-16s' % k[5:]
    for n in names:
        if n in ref['scalars'][k]:
            v, g = float(ref['scalars'][k][n]), float(got['scalars'][k][n])
            row += '%15.2e' % (abs(g - v) / max(abs(v), 1e-30))
        else:
            row += '%15s' % '-'
    print(row)
print("the ORACLE against itself from weights 1e-5 / 3e-5 apart (running max of the relative deviation, every 5th iteration):")
for n in names:
    print('  %-16s' % n + ' '.join('%.0e' % x for x in env[n][::5]))
bad, worst, where, chaotic = cases.compare_trajectories(got, ref, envelope=env)
print('failures', len(bad), 'worst error / allowance', worst, where)
print('outside 1e-3 (1 + it/10) but inside 3 x the envelope:', len(chaotic), sorted(set((c[0][5:8], c[1]) for c in chaotic)))
print('pose rel err', np.abs(got['pose'] - ref['pose']).max() / np.abs(ref['pose']).max(), 'worst joint equal', bool((got['worst_joint'] == ref['worst_joint']).all()),
      'mean err mm', got['mean_err'], ref['mean_err'])
dw = [(float(np.abs(got['dis'][k] - ref['dis'][k]).max()), float(np.abs(ref['dis'][k]).max()), k) for k in ref['dis']]
print('dis weights after the trajectory: max |diff| per tensor', sorted(dw, reverse=True)[:6])
