#!/usr/bin/env python
"""Runs the REAL reference data step (src/data/dataset_hand2.py: normalize, augmentCrop; src/utils/handdetector.py;
src/data/importers.py NYUImporter) on seeded synthetic crops and writes tests/golden/golden_data.npz.

Build container only (needs /root/reference).  cv2 is not installed here: the three cv2 calls on this path
(getRotationMatrix2D, warpAffine, warpPerspective) are served by the restatement in oracle/data_ref.py — see that
file's header: these vectors pin the reference's own arithmetic around the warp, not OpenCV's warp.
usage: python tests/golden/make_golden_data.py
"""
import os
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

import data_cases  # noqa: E402
import ref_shim  # noqa: E402
from oracle import data_ref  # noqa: E402


def cv2_standin():
    cv2 = types.ModuleType('cv2')
    cv2.INTER_NEAREST, cv2.INTER_LINEAR, cv2.BORDER_CONSTANT = 0, 1, 0
    cv2.getRotationMatrix2D = data_ref.get_rotation_matrix_2d

    def warpAffine(src, M, dsize, flags=0, borderMode=0, borderValue=0):
        assert flags == cv2.INTER_NEAREST and tuple(dsize) == (src.shape[1], src.shape[0])
        return data_ref.warp_affine_nn(src, M, float(borderValue))

    def warpPerspective(src, M, dsize, flags=0, borderMode=0, borderValue=0):
        assert flags == cv2.INTER_NEAREST
        return data_ref.warp_perspective_nn(src, M, dsize, float(borderValue))

    cv2.warpAffine, cv2.warpPerspective = warpAffine, warpPerspective
    return cv2


def main():
    ds, HandDetector, NYUImporter = ref_shim.load_reference_data(cv2_standin())
    di = NYUImporter('/nonexistent', useCache=False, refineNet=None, allJoints=True, cacheDir='/nonexistent')
    out = {}
    n_cases = 0
    for set_id, aug_modes in enumerate(data_cases.AUG_SETS):
        for k in range(6 if set_id == 0 else 3):
            seed = 1000 * set_id + k
            s = data_cases.make_sample(seed)
            cube = np.asarray(s['cube'], 'float32')
            com = np.asarray(s['com3D'], 'float32')
            img = np.asarray(s['dpt'].copy(), 'float32')
            img = ds.normalize(img, com, cube)                                        # dataset_hand2.py:341
            hd = HandDetector(img.copy(), abs(di.fx), abs(di.fy), importer=di)         # :309
            com2D = di.joint3DToImg(com)
            M = np.asarray(hd.comToTransform(com2D, cube, (128, 128)), 'float32')     # what cropArea3D stores as T
            rng = np.random.RandomState(seed + 7)
            imgD, _, label, cube_o, com_o, M_o, rot = ds.augmentCrop(img.copy(), s['gt3D'].copy(), com2D, cube, M,
                                                                     list(aug_modes), hd, rng=rng)   # :353-354
            p = 'c%02d.' % n_cases
            out[p + 'seed'] = np.int64(seed)
            out[p + 'set'] = np.int64(set_id)
            out[p + 'M'] = M
            out[p + 'com2D'] = np.asarray(com2D, np.float32)
            out[p + 'norm'] = img
            out[p + 'img'] = np.asarray(imgD, np.float32)
            out[p + 'label'] = np.asarray(label, np.float32)
            out[p + 'cube'] = np.asarray(cube_o, np.float32)
            out[p + 'com_out'] = np.asarray(com_o, np.float32)
            out[p + 'com3D_out'] = np.asarray(di.jointImgTo3D(com_o), np.float32)    # what __getitem__ returns (:364)
            out[p + 'M_out'] = np.asarray(M_o, np.float32)
            out[p + 'rot'] = np.float64(rot)
            n_cases += 1
    out['n_cases'] = np.int64(n_cases)
    out['numpy_version'] = np.array(np.__version__)
    np.savez_compressed(os.path.join(HERE, 'golden_data.npz'), **out)
    print('wrote %d cases' % n_cases)


if __name__ == '__main__':
    main()
