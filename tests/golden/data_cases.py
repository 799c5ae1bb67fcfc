"""Seeded synthetic inputs for the data-step parity cases: raw 128x128 depth crops in mm with a
zero background, the crop's CoM, cube and 36 joints.  Shared by tests/golden/make_golden_data.py (which runs the real
reference on them) and the parity tests (oracle / HIP kernels).  No dataset exists here, so the crops are synthetic:
an elliptical blob of smooth depth around the CoM plus a handful of out-of-cube pixels that exercise the clipping."""
import numpy as np

AUG_SETS = (['none', 'com', 'rot'],          # the list the NYU / ICVL datasets hard-wire (dataset_hand2.py:271)
            ['com'], ['rot'], ['sc'], ['none'], ['com', 'rot', 'sc'])


def make_sample(seed, size=128, cube_mm=300.0):
    rs = np.random.RandomState(seed)
    com3D = np.array([rs.uniform(-80, 80), rs.uniform(-60, 60), rs.uniform(450, 900)], np.float32)
    cube = np.array([cube_mm] * 3, np.float32)
    yy, xx = np.mgrid[0:size, 0:size].astype(np.float32)
    cy, cx = size / 2 + rs.uniform(-5, 5), size / 2 + rs.uniform(-5, 5)
    ry, rx = rs.uniform(0.25, 0.4) * size, rs.uniform(0.2, 0.35) * size
    mask = ((yy - cy) / ry) ** 2 + ((xx - cx) / rx) ** 2 < 1.0
    f = rs.uniform(-1, 1, size=(size // 8, size // 8)).astype(np.float32)
    f = np.kron(f, np.ones((8, 8), np.float32))
    for _ in range(2):
        f = (f + np.roll(f, 1, 0) + np.roll(f, -1, 0)) / 3.0
        f = (f + np.roll(f, 1, 1) + np.roll(f, -1, 1)) / 3.0
    dpt = np.zeros((size, size), np.float32)
    dpt[mask] = (com3D[2] + 110.0 * f)[mask]
    for _ in range(6):                       # a few pixels in front of / behind the cube
        y, x = rs.randint(0, size, 2)
        dpt[y, x] = com3D[2] + rs.choice([-1.0, 1.0]) * rs.uniform(160, 260)
    gt3D = rs.normal(0.0, 40.0, size=(36, 3)).astype(np.float32)
    return dict(dpt=dpt, com3D=com3D, cube=cube, gt3D=gt3D)
