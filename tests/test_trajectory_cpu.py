"""The trajectory harness itself (tests/golden/cases.py: run_trajectory / compare_trajectories) on the CPU oracle: the scheduler
milestone falls inside the trajectory, and two f32 trajectories started 1e-7 apart stay an order of magnitude inside the bound the
GPU test (tests/test_trajectory_gpu.py) holds the HIP trainer to — i.e. that bound measures the implementation, not chaos."""
import numpy as np
import torch

import cases
from oracle import lsps_ref


def test_oracle_trajectory_is_stable_under_a_1e7_perturbation_and_sees_the_milestone():
    torch.set_num_threads(4)
    O = cases.NativeAdapter(lsps_ref, 'cpu', trainer_kwargs=dict(literal=False))
    kw = dict(n=2, n_pre=10, n_est=6, held_out=4, cadence=3)
    a = cases.run_trajectory(O, 'tiny', lsps_ref, **kw)
    # schedulers at 197, stepped at iterations 2, 5, 8: milestone 200 (lr x 0.5, lsps_trainer.py:32-34) takes effect at iteration 8
    assert a['lrs'][7] == (1e-4, 1e-4) and a['lrs'][8] == (5e-5, 5e-5) and a['lrs'][-1] == (5e-5, 5e-5), a['lrs']
    assert all(np.isfinite(float(v)) for it in a['scalars'].values() for v in it.values())
    b = cases.run_trajectory(O, 'tiny', lsps_ref, perturb=1e-7, **kw)
    bad, worst, where = cases.compare_trajectories(b, a, rtol=1e-3, growth=10.0)
    assert not bad and worst < 0.3, (worst, where)
    c = cases.run_trajectory(O, 'tiny', lsps_ref, perturb=3e-2, **kw)                    # and a real difference is seen
    bad, worst, where = cases.compare_trajectories(c, a, rtol=1e-3, growth=10.0)
    assert bad
