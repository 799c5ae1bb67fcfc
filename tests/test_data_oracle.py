"""oracle/data_ref.py against the vectors the REAL reference data step produced (tests/golden/golden_data.npz,
made by tests/golden/make_golden_data.py).  Pins the RNG draw order, the CoM / rotation / scale geometry, the crop
transforms, the z-thresholds, the normalisation tail and the label transforms of dataset_hand2.augmentCrop; the
OpenCV warp is restated, not pinned (see oracle/data_ref.py's header)."""
import os

import numpy as np
import pytest

import data_cases
from oracle import data_ref

G = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'golden_data.npz'))
N_CASES = int(G['n_cases'])


def _oracle_case(i):
    p = 'c%02d.' % i
    seed, set_id = int(G[p + 'seed']), int(G[p + 'set'])
    s = data_cases.make_sample(seed)
    cam = data_ref.Camera(*data_ref.NYU_INTRINSICS)
    det = data_ref.Detector(cam)
    com = np.asarray(s['com3D'], 'float32')
    cube = np.asarray(s['cube'], 'float32')
    norm = data_ref.normalize(s['dpt'].copy(), com, cube)
    com2D = cam.joint3DToImg(com)
    M = np.asarray(det.comToTransform(com2D, cube, (128, 128)), 'float32')
    rng = np.random.RandomState(seed + 7)
    res = data_ref.augment_crop(norm.copy(), s['gt3D'].copy(), com2D, cube, M, list(data_cases.AUG_SETS[set_id]), det, rng)
    return p, norm, com2D, M, res, cam


@pytest.mark.parametrize('i', range(N_CASES))
def test_augment_crop_matches_reference(i):
    p, norm, com2D, M, (imgD, label, cube_o, com_o, M_o, rot), cam = _oracle_case(i)
    assert np.array_equal(norm, G[p + 'norm'])
    assert np.array_equal(com2D, G[p + 'com2D'])
    assert np.array_equal(M, G[p + 'M'])
    assert np.array_equal(np.asarray(imgD, np.float32), G[p + 'img'])
    assert np.array_equal(np.asarray(label, np.float32), G[p + 'label'])
    assert np.array_equal(np.asarray(cube_o, np.float32), G[p + 'cube'])
    assert np.array_equal(np.asarray(com_o, np.float32), G[p + 'com_out'])
    assert np.array_equal(np.asarray(cam.jointImgTo3D(com_o), np.float32), G[p + 'com3D_out'])
    assert np.array_equal(np.asarray(M_o, np.float32), G[p + 'M_out'])
    assert float(rot) == float(G[p + 'rot'])


def test_cases_cover_every_mode():
    """The captured cases really took each branch: images differ from the un-augmented crop for com / rot / sc."""
    moved = {}
    for i in range(N_CASES):
        p = 'c%02d.' % i
        # 'none' still clips the out-of-cube pixels to the cube faces (dataset_hand2.py:111-112)
        moved.setdefault(int(G[p + 'set']), []).append(not np.array_equal(G[p + 'img'], np.clip(G[p + 'norm'], -1, 1)))
    assert all(moved[1]) and all(moved[2]) and all(moved[3])      # forced com, rot, sc
    assert not any(moved[4])                                      # forced none
    assert any(moved[0]) and not all(moved[0])                    # the dataset's list: a mix


def test_warp_restatement_properties():
    """OpenCV restatement sanity (the unpinned part): identity maps are exact copies, a 90-degree rotation about the
    centre is an index permutation, a perspective matrix that is a pure shift moves pixels by whole columns."""
    rs = np.random.RandomState(3)
    a = rs.uniform(1, 9, size=(128, 128)).astype(np.float32)
    assert np.array_equal(data_ref.warp_affine_nn(a, np.array([[1., 0, 0], [0, 1., 0]])), a)
    assert np.array_equal(data_ref.warp_perspective_nn(a, np.eye(3), (128, 128)), a)
    r90 = data_ref.warp_affine_nn(a, data_ref.get_rotation_matrix_2d((64, 64), 90, 1))
    assert np.array_equal(r90[1:, 1:], np.rot90(a, 1)[:-1, 1:]) or np.array_equal(r90[1:, :], np.rot90(a, 1)[:-1, :]) \
        or np.array_equal(r90[:, 1:], np.rot90(a)[:, :-1]) or np.array_equal(r90[1:, 1:], np.rot90(a)[1:, :-1]) \
        or np.array_equal(r90[1:], np.rot90(a)[:-1])
    sh = np.eye(3)
    sh[0, 2] = 5
    w = data_ref.warp_perspective_nn(a, sh, (128, 128))
    assert np.array_equal(w[:, 5:], a[:, :-5]) and not w[:, :5].any()
