"""Data step either side of the depth path, on the GPU.

Mirror of the pieces of the reference's dataset classes that touch every training sample
(``src/data/dataset_hand2.py``): ``normalize`` (:27-31) and ``augmentCrop`` (:34-119) with the ``HandDetector``
geometry behind it (``src/utils/handdetector.py:206-258, 682-805``), split the MI355X way:

  * ``plan_augmentation`` — the per-sample geometry (RNG draws in the reference's order, new CoM / cube / crop
    transform, label transform, inverted warp matrix, z-thresholds).  A few hundred flops per sample of scalar numpy:
    host work, safe to run in DataLoader worker processes.  It keeps the reference's operand types (float32 array
    elements, python-float intrinsics, float64 matrices) so that numpy rounds every step as it does for the reference.
  * ``CropPipeline.normalize`` / ``CropPipeline.augment`` — the per-pixel work for a whole batch in one HIP launch each
    (``lsps_crop_normalize`` / ``lsps_crop_augment`` in ``csrc/data.hip``): crops stay resident in HBM and are read
    and written once.

``load_sequence_cache`` reads the reference's pickle cache of pre-cropped sequences (``src/data/importers.py:1027-1044``,
written with python-2 cPickle) without needing the reference's ``data.basetypes`` module.
"""
import collections
import io
import pickle

import numpy as np

AUG_STRIDE = 16                      # doubles per sample in the kernel's parameter table (include/lsps_hip.h)
KIND = {'none': 0, 'com': 1, 'sc': 1, 'rot': 2}
DEFAULT_AUG_MODES = ('none', 'com', 'rot')           # dataset_hand2.py:144,271

DepthFrame = collections.namedtuple('DepthFrame', ['dpt', 'gtorig', 'gtcrop', 'T', 'gt3Dorig', 'gt3Dcrop', 'com',
                                                   'fileName', 'subSeqName', 'side', 'extraData'])      # basetypes.py:32-34
NamedImgSequence = collections.namedtuple('NamedImgSequence', ['name', 'data', 'config'])              # basetypes.py:35


class Camera(object):
    """Pinhole projection of the importers (importers.py:84-122; NYU flips y, :1260-1298)."""

    def __init__(self, fx, fy, ux, uy, flip_y):
        self.fx, self.fy, self.ux, self.uy, self.flip_y = fx, fy, ux, uy, flip_y

    def to_3d(self, p):
        """(x, y, z) image coordinates -> metric, float32 [3]."""
        out = np.zeros((3,), np.float32)
        out[0] = (p[0] - self.ux) * p[2] / self.fx
        out[1] = ((self.uy - p[1]) if self.flip_y else (p[1] - self.uy)) * p[2] / self.fy
        out[2] = p[2]
        return out

    def to_img(self, p):
        out = np.zeros((3,), np.float32)
        if p[2] == 0.:
            out[0], out[1] = self.ux, self.uy
            return out
        out[0] = p[0] / p[2] * self.fx + self.ux
        out[1] = (self.uy - p[1] / p[2] * self.fy) if self.flip_y else (p[1] / p[2] * self.fy + self.uy)
        out[2] = p[2]
        return out


NYU_CAMERA = Camera(588.03, 587.07, 320., 240., True)        # importers.py:961
ICVL_CAMERA = Camera(241.42, 241.42, 160., 120., False)      # importers.py:203


def crop_bounds(cam, com, size):
    """HandDetector.comToBounds (handdetector.py:206-228), well-defined CoM."""
    fx, fy = abs(cam.fx), abs(cam.fy)                          # dataset_hand2.py:155,310
    zstart, zend = com[2] - size[2] / 2., com[2] + size[2] / 2.
    xs = int(np.floor((com[0] * com[2] / fx - size[0] / 2.) / com[2] * fx + 0.5))
    xe = int(np.floor((com[0] * com[2] / fx + size[0] / 2.) / com[2] * fx + 0.5))
    ys = int(np.floor((com[1] * com[2] / fy - size[1] / 2.) / com[2] * fy + 0.5))
    ye = int(np.floor((com[1] * com[2] / fy + size[1] / 2.) / com[2] * fy + 0.5))
    return xs, xe, ys, ye, zstart, zend


def crop_transform(cam, com, size, dsize=(128, 128)):
    """HandDetector.comToTransform (handdetector.py:230-258): 3x3 float64 map full image -> crop."""
    xs, xe, ys, ye, _, _ = crop_bounds(cam, com, size)
    wb, hb = xe - xs, ye - ys
    if wb > hb:
        f, sz = dsize[0] / float(wb), (dsize[0], hb * dsize[0] / wb)
    else:
        f, sz = dsize[1] / float(hb), (wb * dsize[1] / hb, dsize[1])
    shift = np.array([[1., 0., -xs], [0., 1., -ys], [0., 0., 1.]])
    scale = np.eye(3) * f
    scale[2, 2] = 1
    centre = np.eye(3)
    centre[0, 2] = int(np.floor(dsize[0] / 2. - sz[1] / 2.))
    centre[1, 2] = int(np.floor(dsize[1] / 2. - sz[0] / 2.))
    return np.dot(centre, np.dot(scale, shift))


def _cv_invert3(A):
    """Inverse of a 3x3 double matrix the way cv::invert forms it: cofactors times 1/det, row-major list of 9."""
    S = [[float(A[i][j]) for j in range(3)] for i in range(3)]

    def minor(r0, c0, r1, c1, r2, c2, r3, c3):
        return S[r0][c0] * S[r1][c1] - S[r2][c2] * S[r3][c3]

    det = (S[0][0] * minor(1, 1, 2, 2, 1, 2, 2, 1) - S[0][1] * minor(1, 0, 2, 2, 1, 2, 2, 0)
           + S[0][2] * minor(1, 0, 2, 1, 1, 1, 2, 0))
    if det == 0:
        return [0.0] * 9
    d = 1.0 / det
    return [minor(1, 1, 2, 2, 1, 2, 2, 1) * d, minor(0, 2, 2, 1, 0, 1, 2, 2) * d, minor(0, 1, 1, 2, 0, 2, 1, 1) * d,
            minor(1, 2, 2, 0, 1, 0, 2, 2) * d, minor(0, 0, 2, 2, 0, 2, 2, 0) * d, minor(0, 2, 1, 0, 0, 0, 1, 2) * d,
            minor(1, 0, 2, 1, 1, 1, 2, 0) * d, minor(0, 1, 2, 0, 0, 0, 2, 1) * d, minor(0, 0, 1, 1, 0, 1, 1, 0) * d]


def _cv_rotation_inverse(center, angle_deg):
    """cv2.getRotationMatrix2D(center, angle, 1) followed by the inversion cv2.warpAffine applies to it."""
    a = angle_deg * np.pi / 180.0
    alpha, beta = float(np.cos(a)), float(np.sin(a))
    cx, cy = float(center[0]), float(center[1])
    m = [alpha, beta, (1 - alpha) * cx - beta * cy, -beta, alpha, beta * cx + (1 - alpha) * cy]
    D = m[0] * m[4] - m[1] * m[3]
    D = 1.0 / D if D != 0 else 0.0
    A11, A22 = m[4] * D, m[0] * D
    m[0] = A11
    m[1] *= -D
    m[3] *= -D
    m[4] = A22
    b1 = -m[0] * m[2] - m[1] * m[5]
    b2 = -m[3] * m[2] - m[4] * m[5]
    m[2], m[5] = b1, b2
    return m


def draw_augmentation(rng, n_modes, sigma_com=10., sigma_sc=0.05, rot_range=180.):
    """The four draws of one augmentCrop call, in its order (dataset_hand2.py:69-72)."""
    mode = rng.randint(0, n_modes)
    off = rng.randn(3) * sigma_com
    rot = rng.uniform(-rot_range, rot_range)
    sc = abs(1. + rng.randn() * sigma_sc)
    return mode, off, rot, sc


Plan = collections.namedtuple('Plan', ['prm', 'label', 'cube', 'com', 'com3D', 'M', 'rot', 'mode'])


def plan_augmentation(cam, gt3Dcrop, com, cube, M, aug_modes, rng, dsize=(128, 128), sigma_com=None, sigma_sc=None,
                      rot_range=None):
    """Everything of augmentCrop except the pixels.  `com`: CoM in image coordinates (x, y, z_mm), float32 [3];
    `cube`: float32 [3]; `M`: float32 [3, 3] crop transform; `gt3Dcrop`: float32 [J, 3] joints relative to the CoM.
    Returns a Plan: `prm` (float64 [AUG_STRIDE], one row of the kernel's table), the augmented `label` [J, 3]
    (joints / (cube_z/2)), and the new cube / CoM (image + metric) / crop transform / rotation, as augmentCrop and
    dataset_hand_NYU.__getitem__ (dataset_hand2.py:353-364) return them."""
    sigma_com = 10. if sigma_com is None else sigma_com
    sigma_sc = 0.05 if sigma_sc is None else sigma_sc
    rot_range = 180. if rot_range is None else rot_range
    mode, off, rot, sc = draw_augmentation(rng, len(aug_modes), sigma_com, sigma_sc, rot_range)
    name = aug_modes[mode]
    if name not in KIND:
        raise NotImplementedError(name)
    prm = np.zeros((AUG_STRIDE,), np.float64)
    prm[1], prm[2] = com[2], cube[2] / 2.
    kind = 0
    new_com, new_cube, new_M, joints = com, cube, M, gt3Dcrop
    H, W = int(dsize[1]), int(dsize[0])
    if name == 'com':                                             # HandDetector.moveCoM (handdetector.py:682-714)
        rot = 0.
        if not np.allclose(off, 0.):
            new_com = cam.to_img(cam.to_3d(com) + off)
            if not (np.allclose(com[2], 0.) or np.allclose(new_com[2], 0.)):
                new_M = crop_transform(cam, new_com, cube, (H, W))     # the reference passes dpt.shape (rows, cols)
                fwd = np.dot(new_M, np.linalg.inv(M))
                _, _, _, _, zs, ze = crop_bounds(cam, new_com, cube)
                kind, prm[5], prm[6] = 1, zs, ze
                prm[7:16] = _cv_invert3(fwd)
            joints = gt3Dcrop + cam.to_3d(com) - cam.to_3d(new_com)
    elif name == 'rot':                                           # HandDetector.rotateHand (handdetector.py:716-752)
        if not np.allclose(rot, 0.):
            rot = np.mod(rot, 360)
            kind = 2
            prm[7:13] = _cv_rotation_inverse((W // 2, H // 2), -rot)
            com3D = cam.to_3d(com)
            alpha = rot * np.pi / 180.
            # all joints at once; every elementwise step has the operand types of the per-joint reference code
            # (float32 arrays with python-float intrinsics stay float32; products with the float64 cos / sin are float64
            # and round once when stored), so the result is bit-identical to the joint-by-joint loop
            q = (gt3Dcrop + com3D).astype(np.float32)
            pj = np.zeros_like(q)
            zero = q[:, 2] == 0.
            qz = np.where(zero, np.float32(1), q[:, 2])
            pj[:, 0] = q[:, 0] / qz * cam.fx + cam.ux
            pj[:, 1] = (cam.uy - q[:, 1] / qz * cam.fy) if cam.flip_y else (q[:, 1] / qz * cam.fy + cam.uy)
            pj[:, 2] = q[:, 2]
            pj[zero, 0], pj[zero, 1], pj[zero, 2] = cam.ux, cam.uy, 0.
            pj[:, 0:2] -= com[0:2]
            rj = np.zeros_like(pj)
            rj[:, 0] = pj[:, 0] * np.cos(alpha) - pj[:, 1] * np.sin(alpha)
            rj[:, 1] = pj[:, 0] * np.sin(alpha) + pj[:, 1] * np.cos(alpha)
            rj[:, 2] = pj[:, 2]
            rj[:, 0:2] += com[0:2]
            joints = np.zeros_like(rj)
            joints[:, 0] = (rj[:, 0] - cam.ux) * rj[:, 2] / cam.fx
            joints[:, 1] = ((cam.uy - rj[:, 1]) if cam.flip_y else (rj[:, 1] - cam.uy)) * rj[:, 2] / cam.fy
            joints[:, 2] = rj[:, 2]
            joints = joints - com3D
    elif name == 'sc':                                            # HandDetector.scaleHand (handdetector.py:755-784)
        rot = 0.
        if not np.allclose(sc, 1.):
            new_cube = [s * sc for s in cube]
            if not np.allclose(com[2], 0.):
                new_M = crop_transform(cam, com, new_cube, (H, W))
                fwd = np.dot(new_M, np.linalg.inv(M))
                _, _, _, _, zs, ze = crop_bounds(cam, com, cube)       # thresholds use the OLD cube (:778)
                kind, prm[5], prm[6] = 1, zs, ze
                prm[7:16] = _cv_invert3(fwd)
    else:
        rot = 0.
    label = joints / (new_cube[2] / 2.)
    prm[0], prm[3], prm[4] = kind, new_com[2], new_cube[2] / 2.
    return Plan(prm, np.asarray(label, np.float32), np.asarray(new_cube), new_com, cam.to_3d(new_com),
                np.array(new_M, dtype='float32'), rot, name)


class CropPipeline(object):
    """Batched GPU side.  Tensors are float32 HIP tensors [N, 1, H, W] (or [N, H, W]); fails loudly without the
    HIP library (no CPU fallback)."""

    def __init__(self, device):
        import torch
        from . import _lib
        self.torch, self._lib, self.device = torch, _lib, torch.device(device)
        self.L = _lib.lib()

    def normalize(self, dpt, com_z, cube_z, out=None):
        """dataset_hand2.normalize for a batch: `dpt` raw depth crops in mm (0 = no measurement), `com_z` / `cube_z`
        per-sample [N] (host arrays or tensors)."""
        t = self.torch
        dpt = dpt.contiguous()
        N, HW = dpt.shape[0], dpt[0].numel()
        cz = t.as_tensor(np.asarray(com_z, np.float32) if not t.is_tensor(com_z) else com_z, dtype=t.float32).to(self.device)
        hf = t.as_tensor(np.asarray(cube_z, np.float32) if not t.is_tensor(cube_z) else cube_z, dtype=t.float32).to(self.device) / 2.0
        out = t.empty_like(dpt) if out is None else out
        self._lib.check(self.L.lsps_crop_normalize(self._lib.ptr(dpt), self._lib.ptr(cz.contiguous()), self._lib.ptr(hf.contiguous()),
                                                   self._lib.ptr(out), N, HW, self._lib.stream()), 'crop_normalize')
        return out

    def augment(self, x, plans, out=None):
        """The pixel part of augmentCrop for a batch of NORMALISED crops `x`; `plans`: list of Plan (one per sample)
        or a float64 array [N, AUG_STRIDE]."""
        t = self.torch
        x = x.contiguous()
        N, H, W = x.shape[0], x.shape[-2], x.shape[-1]
        table = np.stack([p.prm for p in plans]) if not isinstance(plans, np.ndarray) else plans
        assert table.shape == (N, AUG_STRIDE) and table.dtype == np.float64
        prm = t.from_numpy(np.ascontiguousarray(table)).to(self.device)
        out = t.empty_like(x) if out is None else out
        self._lib.check(self.L.lsps_crop_augment(self._lib.ptr(x), self._lib.ptr(prm, t.float64), self._lib.ptr(out), N, H, W,
                                                 self._lib.stream()), 'crop_augment')
        return out


class _CacheUnpickler(pickle.Unpickler):
    """Maps the reference's `data.basetypes` records onto the namedtuples above.  Only the globals a sequence cache
    legitimately contains are resolved (numpy array reconstruction + the two records); anything else raises, so a cache
    file from an untrusted source cannot name arbitrary callables (the reference's cPickle.load would run them)."""

    _ALLOWED = {
        ('numpy.core.multiarray', '_reconstruct'), ('numpy._core.multiarray', '_reconstruct'),
        ('numpy.core.multiarray', 'scalar'), ('numpy._core.multiarray', 'scalar'),
        ('numpy', 'ndarray'), ('numpy', 'dtype'),
        ('collections', 'OrderedDict'), ('__builtin__', 'tuple'), ('__builtin__', 'list'), ('__builtin__', 'dict'),
        ('builtins', 'tuple'), ('builtins', 'list'), ('builtins', 'dict'),
        ('_codecs', 'encode'),             # how a python-3 writer stores the array bytes in protocol 2 (str -> latin1)
    }

    def find_class(self, module, name):
        if module.endswith('basetypes') and name == 'DepthFrame':
            return DepthFrame
        if module.endswith('basetypes') and name == 'NamedImgSequence':
            return NamedImgSequence
        if (module, name) in self._ALLOWED:
            return super(_CacheUnpickler, self).find_class(module, name)
        raise pickle.UnpicklingError("sequence cache names a global outside the whitelist: %s.%s" % (module, name))


def cache_file_name(cache_dir, importer_name, seq_name, hand=None, all_joints=True, crop_joint_idx=32, docom=False,
                    refine=False, cube0=300):
    """importers.py:1027-1029: '<cacheDir>/<Importer>_<seq>_<hand>_<allJoints>_<cropJoint>_<gt|com|comref>_<cube0>__cache.pkl'."""
    mode = 'gt' if not docom and not refine else ('com' if docom and not refine else 'comref')      # handdetector.py:74-91
    return '{}/{}_{}_{}_{}_{}_{}_{}__cache.pkl'.format(cache_dir, importer_name, seq_name, hand, all_joints, crop_joint_idx,
                                                       mode, cube0)


def load_sequence_cache(path_or_bytes, shuffle_rng=None, nmax=None):
    """Reads `(seqName, data, config)` as importers.py:1031-1044 does: optional in-place shuffle with the dataset's
    RandomState, optional truncation.  Returns a NamedImgSequence of DepthFrame records."""
    if isinstance(path_or_bytes, (bytes, bytearray)):
        f = io.BytesIO(path_or_bytes)
    else:
        f = open(path_or_bytes, 'rb')
    try:
        seq_name, data, config = _CacheUnpickler(f, encoding='latin1').load()
    finally:
        f.close()
    if shuffle_rng is not None:
        shuffle_rng.shuffle(data)
    if nmax is not None and not np.isinf(nmax):
        data = data[0:int(nmax)]
    return NamedImgSequence(seq_name, data, config)
