"""Host side of the data step (lsps_amd/data.py): the augmentation planner against the vectors captured from the
real reference (everything augmentCrop returns except the pixels), and the sequence-cache reader."""
import os
import pickle
import sys
import types

import numpy as np
import pytest

import data_cases
from lsps_amd import data as ldata

G = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'golden_data.npz'))
N_CASES = int(G['n_cases'])


def plan_case(i):
    p = 'c%02d.' % i
    seed, set_id = int(G[p + 'seed']), int(G[p + 'set'])
    s = data_cases.make_sample(seed)
    cam = ldata.NYU_CAMERA
    com2D = cam.to_img(np.asarray(s['com3D'], 'float32'))
    M = np.asarray(ldata.crop_transform(cam, com2D, s['cube'], (128, 128)), 'float32')
    rng = np.random.RandomState(seed + 7)
    plan = ldata.plan_augmentation(cam, s['gt3D'].copy(), com2D, s['cube'], M, list(data_cases.AUG_SETS[set_id]), rng)
    return p, s, com2D, M, plan


@pytest.mark.parametrize('i', range(N_CASES))
def test_plan_matches_reference(i):
    p, s, com2D, M, plan = plan_case(i)
    assert np.array_equal(com2D, G[p + 'com2D'])
    assert np.array_equal(M, G[p + 'M'])
    assert np.array_equal(plan.label, G[p + 'label'])
    assert np.array_equal(np.asarray(plan.cube, np.float32), G[p + 'cube'])
    assert np.array_equal(np.asarray(plan.com, np.float32), G[p + 'com_out'])
    assert np.array_equal(plan.com3D, G[p + 'com3D_out'])
    assert np.array_equal(plan.M, G[p + 'M_out'])
    assert float(plan.rot) == float(G[p + 'rot'])
    assert plan.prm.shape == (ldata.AUG_STRIDE,) and plan.prm.dtype == np.float64


def test_draw_order_is_the_references():
    """mode, 3 x randn, uniform, randn — the RandomState is left where augmentCrop leaves it."""
    a, b = np.random.RandomState(5), np.random.RandomState(5)
    mode, off, rot, sc = ldata.draw_augmentation(a, 3)
    assert mode == b.randint(0, 3)
    assert np.array_equal(off, b.randn(3) * 10.)
    assert rot == b.uniform(-180., 180.)
    assert sc == abs(1. + b.randn() * 0.05)
    assert a.randint(1 << 30) == b.randint(1 << 30)


def test_sequence_cache_reader(tmp_path):
    """A cache written the way the reference writes it (cPickle protocol 2 of (seqName, [DepthFrame...], config) with
    the records' class living in `data.basetypes`, importers.py:1140-1143) loads without that module."""
    import collections
    mod_data, mod_bt = types.ModuleType('data'), types.ModuleType('data.basetypes')
    DF = collections.namedtuple('DepthFrame', ldata.DepthFrame._fields)
    DF.__module__ = 'data.basetypes'
    mod_bt.DepthFrame = DF
    mod_data.basetypes = mod_bt
    saved = {k: sys.modules.get(k) for k in ('data', 'data.basetypes')}
    sys.modules['data'], sys.modules['data.basetypes'] = mod_data, mod_bt
    try:
        frames = []
        for k in range(5):
            s = data_cases.make_sample(k)
            frames.append(DF(s['dpt'], None, None, np.eye(3, dtype=np.float32), None, s['gt3D'], s['com3D'],
                             'depth_1_%07d.png' % (k + 1), '', 'right', {}))
        blob = pickle.dumps(('train', frames, {'cube': (300, 300, 300)}), protocol=2)
    finally:
        for k, v in saved.items():
            if v is None:
                sys.modules.pop(k, None)
            else:
                sys.modules[k] = v
    path = ldata.cache_file_name(str(tmp_path), 'NYUImporter', 'train', None, True, 32, False, False, 300)
    assert os.path.basename(path) == 'NYUImporter_train_None_True_32_gt_300__cache.pkl'
    with open(path, 'wb') as f:
        f.write(blob)
    seq = ldata.load_sequence_cache(path)
    assert seq.name == 'train' and seq.config['cube'] == (300, 300, 300) and len(seq.data) == 5
    assert isinstance(seq.data[0], ldata.DepthFrame) and seq.data[3].fileName == 'depth_1_0000004.png'
    assert np.array_equal(seq.data[2].dpt, data_cases.make_sample(2)['dpt'])
    # shuffle + Nmax as loadSequence applies them (importers.py:1036-1044)
    rng = np.random.RandomState(23455)
    order = list(range(5))
    np.random.RandomState(23455).shuffle(order)
    seq2 = ldata.load_sequence_cache(path, shuffle_rng=rng, nmax=3)
    assert [d.fileName for d in seq2.data] == ['depth_1_%07d.png' % (k + 1) for k in order[:3]]


def test_sequence_cache_reader_rejects_foreign_globals():
    """A cache file is a pickle: the reader resolves numpy reconstruction + the two record types and nothing else."""
    import pickle
    import pytest
    evil = pickle.dumps((os.system, 'echo pwned'), protocol=2)
    with pytest.raises(pickle.UnpicklingError):
        ldata.load_sequence_cache(evil)
