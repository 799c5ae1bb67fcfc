"""Data-step kernels (lsps_crop_normalize / lsps_crop_augment through lsps_amd/data.py) on the GPU: bit-exact against
the vectors captured from the real reference, against the oracle on further seeded batches with every mode, and
size-independent properties at the bench batch size."""
import os

import numpy as np
import pytest
import torch

import data_cases
from lsps_amd import data as ldata
from oracle import data_ref

pytestmark = pytest.mark.gpu
G = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'golden_data.npz'))
N_CASES = int(G['n_cases'])


@pytest.fixture(scope='module')
def pipe():
    if not torch.cuda.is_available():
        pytest.skip("no HIP device")
    return ldata.CropPipeline('cuda:0')


def _inputs(seed):
    s = data_cases.make_sample(seed)
    cam = ldata.NYU_CAMERA
    com2D = cam.to_img(np.asarray(s['com3D'], 'float32'))
    M = np.asarray(ldata.crop_transform(cam, com2D, s['cube'], (128, 128)), 'float32')
    return s, com2D, M


def test_golden_cases_bit_exact(pipe):
    """All captured cases as ONE batch (mixed kinds in one launch)."""
    dpts, comz, cubez, plans = [], [], [], []
    for i in range(N_CASES):
        p = 'c%02d.' % i
        seed, set_id = int(G[p + 'seed']), int(G[p + 'set'])
        s, com2D, M = _inputs(seed)
        rng = np.random.RandomState(seed + 7)
        plans.append(ldata.plan_augmentation(ldata.NYU_CAMERA, s['gt3D'].copy(), com2D, s['cube'], M,
                                             list(data_cases.AUG_SETS[set_id]), rng))
        dpts.append(s['dpt'])
        comz.append(s['com3D'][2])
        cubez.append(s['cube'][2])
    x = torch.from_numpy(np.stack(dpts)[:, None]).cuda()
    norm = pipe.normalize(x, np.array(comz, np.float32), np.array(cubez, np.float32))
    out = pipe.augment(norm, plans)
    norm_h, out_h = norm.cpu().numpy(), out.cpu().numpy()
    for i in range(N_CASES):
        p = 'c%02d.' % i
        assert np.array_equal(norm_h[i, 0], G[p + 'norm']), 'normalize, case %d' % i
        assert np.array_equal(out_h[i, 0], G[p + 'img']), 'augment, case %d (%s)' % (i, plans[i].mode)


@pytest.mark.parametrize('modes', [['none', 'com', 'rot'], ['com', 'rot', 'sc']])
def test_batch_vs_oracle(pipe, modes):
    """48 further samples per mode list, oracle per-sample vs one HIP launch; ICVL camera for half of them."""
    N = 48
    dpts, plans, want, comz, cubez = [], [], [], [], []
    for k in range(N):
        seed = 5000 + k
        s = data_cases.make_sample(seed, cube_mm=300.0 if k % 2 == 0 else 250.0)
        cam_p = ldata.NYU_CAMERA if k % 2 == 0 else ldata.ICVL_CAMERA
        cam_o = data_ref.Camera(*data_ref.NYU_INTRINSICS, flip_y=True) if k % 2 == 0 else \
            data_ref.Camera(*data_ref.ICVL_INTRINSICS, flip_y=False)
        det = data_ref.Detector(cam_o)
        com = np.asarray(s['com3D'], 'float32')
        com2D = cam_o.joint3DToImg(com)
        M = np.asarray(det.comToTransform(com2D, s['cube'], (128, 128)), 'float32')
        norm = data_ref.normalize(s['dpt'].copy(), com, s['cube'])
        img, label, cube_o, com_o, M_o, rot = data_ref.augment_crop(norm.copy(), s['gt3D'].copy(), com2D, s['cube'], M, modes,
                                                                    det, np.random.RandomState(seed))
        plan = ldata.plan_augmentation(cam_p, s['gt3D'].copy(), cam_p.to_img(com), s['cube'], M, modes,
                                       np.random.RandomState(seed))
        assert np.array_equal(plan.label, np.asarray(label, np.float32)) and np.array_equal(plan.M, M_o)
        dpts.append(s['dpt'])
        comz.append(com[2])
        cubez.append(s['cube'][2])
        plans.append(plan)
        want.append(np.asarray(img, np.float32))
    x = torch.from_numpy(np.stack(dpts)[:, None]).cuda()
    out = pipe.augment(pipe.normalize(x, np.array(comz, np.float32), np.array(cubez, np.float32)), plans).cpu().numpy()
    kinds = set()
    for k in range(N):
        assert np.array_equal(out[k, 0], want[k]), 'sample %d mode %s' % (k, plans[k].mode)
        kinds.add(plans[k].mode)
    assert kinds == set(modes)


def test_full_batch_properties(pipe):
    """bs = 256 (BASELINE config 5's per-GPU batch): identity plans reproduce clip(normalize(x)); a rotation by 180
    degrees applied twice is the identity away from the border; outputs stay in [-1, 1] with background exactly +1;
    the kernels are deterministic."""
    N = 256
    rs = np.random.RandomState(11)
    base = [data_cases.make_sample(9000 + k) for k in range(8)]
    dpt = np.stack([base[k % 8]['dpt'] for k in range(N)])[:, None]
    comz = np.array([base[k % 8]['com3D'][2] for k in range(N)], np.float32)
    cubez = np.full((N,), 300.0, np.float32)
    x = pipe.normalize(torch.from_numpy(dpt).cuda(), comz, cubez)
    ident = np.zeros((N, ldata.AUG_STRIDE), np.float64)
    ident[:, 1], ident[:, 3] = comz, comz
    ident[:, 2], ident[:, 4] = 150.0, 150.0
    y0 = pipe.augment(x, ident)
    assert torch.equal(y0, x.clamp(-1, 1))
    rot = ident.copy()
    rot[:, 0] = 2
    rot[:, 7:13] = np.array(ldata._cv_rotation_inverse((64, 64), 180.0))
    y1 = pipe.augment(x, rot)
    y2 = pipe.augment(y1, rot)
    assert torch.equal(y2[..., 2:-2, 2:-2], y0[..., 2:-2, 2:-2])
    assert float(y1.max()) == 1.0 and float(y1.min()) >= -1.0
    assert torch.equal(y1, pipe.augment(x, rot))
    # mixed random plans: every output pixel is either background or a value of the source crop's range
    plans = []
    for k in range(N):
        s = base[k % 8]
        com2D = ldata.NYU_CAMERA.to_img(s['com3D'])
        M = np.asarray(ldata.crop_transform(ldata.NYU_CAMERA, com2D, s['cube'], (128, 128)), 'float32')
        plans.append(ldata.plan_augmentation(ldata.NYU_CAMERA, s['gt3D'], com2D, s['cube'], M, ['none', 'com', 'rot'], rs))
    y = pipe.augment(x, plans)
    assert bool(torch.isfinite(y).all()) and float(y.max()) == 1.0 and float(y.min()) >= -1.0
    frac_bg = float((y == 1.0).float().mean())
    assert 0.4 < frac_bg < 0.95


def test_bad_arguments_fail_loudly(pipe):
    x = torch.zeros(2, 1, 128, 128, device='cuda')
    with pytest.raises(RuntimeError):
        pipe.augment(x, np.zeros((2, ldata.AUG_STRIDE)), out=x)         # in place is refused
    with pytest.raises(Exception):
        pipe.normalize(torch.zeros(2, 1, 5, 5, device='cuda'), [1., 1.], [2., 2.])   # HW % 4 != 0
