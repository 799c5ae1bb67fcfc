"""ORACLE — test infrastructure only.  NOT part of the product path.

CPU restatement (numpy, per-sample loops like the reference) of the data step that feeds the depth path
(SURVEY.md §8(f) row N4): ``normalize`` and ``augmentCrop`` of ``src/data/dataset_hand2.py:27-119`` with the
``HandDetector`` geometry they call (``src/utils/handdetector.py:206-258, 682-805``) and the importer's camera
projection (``src/data/importers.py:84-122`` and the NYU override ``:1260-1298``, intrinsics ``:960``).

Who may import this file: ``tests/`` only.  ``lsps_amd`` never imports it.

Parity pin — TWO parts, stated separately:
  * everything the reference itself computes (RNG draw order, CoM / rotation / scale geometry, crop transforms,
    z-thresholds, the pre-max / zero / clip / normalise tail, the label transforms) is PINNED: the real
    ``augmentCrop`` / ``HandDetector`` are imported in the build container by ``tests/golden/ref_shim.py`` and run by
    ``tests/golden/make_golden_data.py``; ``tests/test_data_oracle.py`` checks this file against those vectors.
  * the nearest-neighbour warps themselves live in a THIRD-PARTY dependency that is absent from /root/reference
    and from this image: OpenCV (``cv2.warpAffine`` / ``cv2.warpPerspective`` / ``cv2.getRotationMatrix2D``; the
    reference pins no version — python-2 era, OpenCV 3.x).  ``warp_affine_nn`` / ``warp_perspective_nn`` /
    ``get_rotation_matrix_2d`` below restate OpenCV's published algorithm (modules/imgproc/src/imgwarp.cpp: the
    inverse map is formed in double; warpAffine uses 10-bit fixed point with round-half offsets, warpPerspective
    rounds the double quotient with cvRound; both in 16x64-pixel blocks; remap NN with BORDER_CONSTANT).  That
    part is PARITY UNPINNED: no cv2 here to generate vectors, and the reference holds none.  The golden capture
    injects these same functions as its ``cv2`` stand-in, so the vectors pin the reference's code AROUND the warp,
    not the warp.

Scalar arithmetic: every expression keeps the reference's operand types (float32 array elements, python-float
intrinsics, float64 matrices), so numpy's promotion rules — those of the numpy this runs under, the same numpy the
golden capture ran the reference under — decide the precision of each step exactly as they do for the reference.
"""
import numpy as np

INT_MIN, INT_MAX = -2147483648, 2147483647
NYU_INTRINSICS = (588.03, 587.07, 320., 240.)      # fx, fy, ux, uy  (importers.py:960)
ICVL_INTRINSICS = (241.42, 241.42, 160., 120.)     # (importers.py:137)


# ----------------------------------------------------------------------------------------------
# OpenCV restatement (third-party, unpinned — see header)
# ----------------------------------------------------------------------------------------------
def cv_round(v):
    """cvRound: round half to even (lrint), saturated to int."""
    v = min(max(v, float(INT_MIN)), float(INT_MAX))
    return int(np.rint(v))


def _sat_short(i):
    return -32768 if i < -32768 else (32767 if i > 32767 else i)


def get_rotation_matrix_2d(center, angle, scale):
    a = angle * np.pi / 180.0
    alpha, beta = np.cos(a) * scale, np.sin(a) * scale
    cx, cy = float(center[0]), float(center[1])
    return np.array([[alpha, beta, (1 - alpha) * cx - beta * cy],
                     [-beta, alpha, beta * cx + (1 - alpha) * cy]], np.float64)


def invert_affine(M):
    """The in-place inversion warpAffine applies to a forward 2x3 matrix."""
    m = [float(v) for v in np.asarray(M, np.float64).reshape(6)]
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


def warp_affine_nn(src, M, border=0.0):
    """cv2.warpAffine(src, M, (W, H), flags=INTER_NEAREST, borderMode=BORDER_CONSTANT, borderValue=border)."""
    H, W = src.shape
    m = invert_affine(M)
    AB_BITS, AB_SCALE = 10, 1024
    rd = AB_SCALE // 2
    dst = np.full((H, W), border, src.dtype)
    adelta = [cv_round(m[0] * x * AB_SCALE) for x in range(W)]
    bdelta = [cv_round(m[3] * x * AB_SCALE) for x in range(W)]
    for y in range(H):
        X0 = cv_round((m[1] * y + m[2]) * AB_SCALE) + rd
        Y0 = cv_round((m[4] * y + m[5]) * AB_SCALE) + rd
        for x in range(W):
            X = _sat_short((X0 + adelta[x]) >> AB_BITS)
            Y = _sat_short((Y0 + bdelta[x]) >> AB_BITS)
            if 0 <= X < W and 0 <= Y < H:
                dst[y, x] = src[Y, X]
    return dst


def invert_3x3(A):
    """cv::invert of a 3x3 double matrix (cofactor formula)."""
    S = np.asarray(A, np.float64)
    d = (S[0, 0] * (S[1, 1] * S[2, 2] - S[1, 2] * S[2, 1]) - S[0, 1] * (S[1, 0] * S[2, 2] - S[1, 2] * S[2, 0])
         + S[0, 2] * (S[1, 0] * S[2, 1] - S[1, 1] * S[2, 0]))
    if d == 0:
        return np.zeros((3, 3), np.float64)
    d = 1.0 / d
    t = np.empty((3, 3), np.float64)
    t[0, 0] = (S[1, 1] * S[2, 2] - S[1, 2] * S[2, 1]) * d
    t[0, 1] = (S[0, 2] * S[2, 1] - S[0, 1] * S[2, 2]) * d
    t[0, 2] = (S[0, 1] * S[1, 2] - S[0, 2] * S[1, 1]) * d
    t[1, 0] = (S[1, 2] * S[2, 0] - S[1, 0] * S[2, 2]) * d
    t[1, 1] = (S[0, 0] * S[2, 2] - S[0, 2] * S[2, 0]) * d
    t[1, 2] = (S[0, 2] * S[1, 0] - S[0, 0] * S[1, 2]) * d
    t[2, 0] = (S[1, 0] * S[2, 1] - S[1, 1] * S[2, 0]) * d
    t[2, 1] = (S[0, 1] * S[2, 0] - S[0, 0] * S[2, 1]) * d
    t[2, 2] = (S[0, 0] * S[1, 1] - S[0, 1] * S[1, 0]) * d
    return t


def perspective_blocks(H, W):
    """Block shape warpPerspective walks the destination in (BLOCK_SZ = 32)."""
    bh0 = min(16, H)
    bw0 = min(1024 // bh0, W)
    bh0 = min(1024 // bw0, H)
    return bh0, bw0


def warp_perspective_nn(src, M, dsize, border=0.0):
    """cv2.warpPerspective(src, M, dsize, flags=INTER_NEAREST, borderMode=BORDER_CONSTANT, borderValue=border)."""
    Hs, Ws = src.shape
    W, H = int(dsize[0]), int(dsize[1])
    m = invert_3x3(M).reshape(9)
    m = [float(v) for v in m]
    bh0, bw0 = perspective_blocks(H, W)
    dst = np.full((H, W), border, src.dtype)
    for by in range(0, H, bh0):
        for bx in range(0, W, bw0):
            bw, bh = min(bw0, W - bx), min(bh0, H - by)
            for y1 in range(bh):
                X0 = m[0] * bx + m[1] * (by + y1) + m[2]
                Y0 = m[3] * bx + m[4] * (by + y1) + m[5]
                W0 = m[6] * bx + m[7] * (by + y1) + m[8]
                for x1 in range(bw):
                    w = W0 + m[6] * x1
                    w = 1.0 / w if w else 0.0
                    fX = max(float(INT_MIN), min(float(INT_MAX), (X0 + m[0] * x1) * w))
                    fY = max(float(INT_MIN), min(float(INT_MAX), (Y0 + m[3] * x1) * w))
                    X, Y = _sat_short(cv_round(fX)), _sat_short(cv_round(fY))
                    if 0 <= X < Ws and 0 <= Y < Hs:
                        dst[by + y1, bx + x1] = src[Y, X]
    return dst


# ----------------------------------------------------------------------------------------------
# importer projection (importers.py:84-122)
# ----------------------------------------------------------------------------------------------
class Camera(object):
    """`flip_y=True` is the NYU importer's convention (image y grows downwards, metric y upwards:
    importers.py:1260-1298); False is the base class / ICVL one (importers.py:84-122)."""

    def __init__(self, fx, fy, ux, uy, flip_y=True):
        self.fx, self.fy, self.ux, self.uy, self.flip_y = fx, fy, ux, uy, flip_y

    def jointImgTo3D(self, s):
        ret = np.zeros((3,), np.float32)
        ret[0] = (s[0] - self.ux) * s[2] / self.fx
        ret[1] = (self.uy - s[1]) * s[2] / self.fy if self.flip_y else (s[1] - self.uy) * s[2] / self.fy
        ret[2] = s[2]
        return ret

    def joint3DToImg(self, s):
        ret = np.zeros((3,), np.float32)
        if s[2] == 0.:
            ret[0], ret[1] = self.ux, self.uy
            return ret
        ret[0] = s[0] / s[2] * self.fx + self.ux
        ret[1] = self.uy - s[1] / s[2] * self.fy if self.flip_y else s[1] / s[2] * self.fy + self.uy
        ret[2] = s[2]
        return ret

    def jointsImgTo3D(self, a):
        return np.stack([self.jointImgTo3D(r) for r in a]).astype(np.float32)

    def joints3DToImg(self, a):
        return np.stack([self.joint3DToImg(r) for r in a]).astype(np.float32)


# ----------------------------------------------------------------------------------------------
# HandDetector geometry (handdetector.py)
# ----------------------------------------------------------------------------------------------
class Detector(object):
    def __init__(self, camera, fx=None, fy=None):
        self.cam = camera
        self.fx = abs(camera.fx if fx is None else fx)          # dataset_hand2.py:309: abs(di.fx), abs(di.fy)
        self.fy = abs(camera.fy if fy is None else fy)

    def comToBounds(self, com, size):                            # handdetector.py:206-228 (well-defined CoM branch)
        zstart = com[2] - size[2] / 2.
        zend = com[2] + size[2] / 2.
        xstart = int(np.floor((com[0] * com[2] / self.fx - size[0] / 2.) / com[2] * self.fx + 0.5))
        xend = int(np.floor((com[0] * com[2] / self.fx + size[0] / 2.) / com[2] * self.fx + 0.5))
        ystart = int(np.floor((com[1] * com[2] / self.fy - size[1] / 2.) / com[2] * self.fy + 0.5))
        yend = int(np.floor((com[1] * com[2] / self.fy + size[1] / 2.) / com[2] * self.fy + 0.5))
        return xstart, xend, ystart, yend, zstart, zend

    def comToTransform(self, com, size, dsize=(128, 128)):       # handdetector.py:230-258
        xstart, xend, ystart, yend, _, _ = self.comToBounds(com, size)
        trans = np.eye(3)
        trans[0, 2] = -xstart
        trans[1, 2] = -ystart
        wb, hb = (xend - xstart), (yend - ystart)
        if wb > hb:
            scale = np.eye(3) * dsize[0] / float(wb)
            sz = (dsize[0], hb * dsize[0] / wb)
        else:
            scale = np.eye(3) * dsize[1] / float(hb)
            sz = (wb * dsize[1] / hb, dsize[1])
        scale[2, 2] = 1
        off = np.eye(3)
        off[0, 2] = int(np.floor(dsize[0] / 2. - sz[1] / 2.))
        off[1, 2] = int(np.floor(dsize[1] / 2. - sz[0] / 2.))
        return np.dot(off, np.dot(scale, trans))

    def recropHand(self, crop, M, Mnew, target_size, background_value=0., nv_val=0., thresh_z=True, com=None,
                   size=(250, 250, 250)):                        # handdetector.py:786-805 (RESIZE_CV2_NN, :71)
        warped = warp_perspective_nn(crop, np.dot(M, Mnew), target_size, float(background_value))
        warped[np.isclose(warped, nv_val)] = background_value
        if thresh_z is True:
            _, _, _, _, zstart, zend = self.comToBounds(com, size)
            msk1 = np.logical_and(warped < zstart, warped != 0)
            msk2 = np.logical_and(warped > zend, warped != 0)
            warped[msk1] = zstart
            warped[msk2] = 0.
        return warped

    def moveCoM(self, dpt, cube, com, off, joints3D, M, pad_value=0):       # handdetector.py:682-714
        if np.allclose(off, 0.):
            return dpt, joints3D, com, M
        new_com = self.cam.joint3DToImg(self.cam.jointImgTo3D(com) + off)
        if not (np.allclose(com[2], 0.) or np.allclose(new_com[2], 0.)):
            Mnew = self.comToTransform(new_com, cube, dpt.shape)
            new_dpt = self.recropHand(dpt, Mnew, np.linalg.inv(M), dpt.shape, background_value=pad_value,
                                      nv_val=32000., thresh_z=True, com=new_com, size=cube)
        else:
            Mnew, new_dpt = M, dpt
        new_joints3D = joints3D + self.cam.jointImgTo3D(com) - self.cam.jointImgTo3D(new_com)
        return new_dpt, new_joints3D, new_com, Mnew

    def rotateHand(self, dpt, cube, com, rot, joints3D, pad_value=0):       # handdetector.py:716-752
        if np.allclose(rot, 0.):
            return dpt, joints3D, rot
        rot = np.mod(rot, 360)
        M = get_rotation_matrix_2d((dpt.shape[1] // 2, dpt.shape[0] // 2), -rot, 1)
        new_dpt = warp_affine_nn(dpt, M, float(pad_value))
        com3D = self.cam.jointImgTo3D(com)
        joint_2D = self.cam.joints3DToImg(joints3D + com3D)
        data_2D = np.zeros_like(joint_2D)
        for k in range(data_2D.shape[0]):
            data_2D[k] = rotate_point_2d(joint_2D[k], com[0:2], rot)
        new_joints3D = (self.cam.jointsImgTo3D(data_2D) - com3D)
        return new_dpt, new_joints3D, rot

    def scaleHand(self, dpt, cube, com, sc, joints3D, M, pad_value=0):      # handdetector.py:755-784
        if np.allclose(sc, 1.):
            return dpt, joints3D, cube, M
        new_cube = [s * sc for s in cube]
        if not np.allclose(com[2], 0.):
            Mnew = self.comToTransform(com, new_cube, dpt.shape)
            new_dpt = self.recropHand(dpt, Mnew, np.linalg.inv(M), dpt.shape, background_value=pad_value,
                                      nv_val=32000., thresh_z=True, com=com, size=cube)
        else:
            Mnew, new_dpt = M, dpt
        return new_dpt, joints3D, new_cube, Mnew


def rotate_point_2d(p1, center, angle):                # transformations.py:71-88
    alpha = angle * np.pi / 180.
    pp = p1.copy()
    pp[0:2] -= center[0:2]
    pr = np.zeros_like(pp)
    pr[0] = pp[0] * np.cos(alpha) - pp[1] * np.sin(alpha)
    pr[1] = pp[0] * np.sin(alpha) + pp[1] * np.cos(alpha)
    pr[2] = pp[2]
    pr[0:2] += center[0:2]
    return pr


# ----------------------------------------------------------------------------------------------
# dataset_hand2.py
# ----------------------------------------------------------------------------------------------
def normalize(img, com, cube):                         # dataset_hand2.py:27-31 (in place on a float32 array)
    img[img == 0] = com[2] + (cube[2] / 2.)
    img -= com[2]
    img /= (cube[2] / 2.)
    return img


def draw(rng, n_modes, sigma_com=10., sigma_sc=0.05, rot_range=180.):
    """The four RandomState draws of one augmentCrop call, in the reference's order (dataset_hand2.py:69-72)."""
    mode = rng.randint(0, n_modes)
    off = rng.randn(3) * sigma_com
    rot = rng.uniform(-rot_range, rot_range)
    sc = abs(1. + rng.randn() * sigma_sc)
    return mode, off, rot, sc


def augment_crop(img, gt3Dcrop, com, cube, M, aug_modes, det, rng, sigma_com=10., sigma_sc=0.05, rot_range=180.):
    """dataset_hand2.py:34-119 with normZeroOne=False.  `img`: normalised float32 [H, W]; `com`: image coords (x, y, z)
    float32; `cube`: float32 [3]; `M`: float32 [3, 3] crop transform.  Returns (imgD, curLabel, cube, com, M, rot)."""
    img = img * (cube[2] / 2.) + com[2]
    premax = img.max()
    mode, off, rot, sc = draw(rng, len(aug_modes), sigma_com, sigma_sc, rot_range)
    name = aug_modes[mode]
    if name == 'com':
        rot, sc = 0., 1.
        imgD, new_joints3D, com, M = det.moveCoM(img.astype('float32'), cube, com, off, gt3Dcrop, M, pad_value=0)
        curLabel = new_joints3D / (cube[2] / 2.)
    elif name == 'rot':
        imgD, new_joints3D, rot = det.rotateHand(img.astype('float32'), cube, com, rot, gt3Dcrop, pad_value=0)
        curLabel = new_joints3D / (cube[2] / 2.)
    elif name == 'sc':
        rot = 0.
        imgD, new_joints3D, cube, M = det.scaleHand(img.astype('float32'), cube, com, sc, gt3Dcrop, M, pad_value=0)
        curLabel = new_joints3D / (cube[2] / 2.)
    elif name == 'none':
        rot = 0.
        imgD = img
        curLabel = gt3Dcrop / (cube[2] / 2.)
    else:
        raise NotImplementedError()
    imgD[imgD == premax] = com[2] + (cube[2] / 2.)
    imgD[imgD == 0] = com[2] + (cube[2] / 2.)
    imgD[imgD >= com[2] + (cube[2] / 2.)] = com[2] + (cube[2] / 2.)
    imgD[imgD <= com[2] - (cube[2] / 2.)] = com[2] - (cube[2] / 2.)
    imgD -= com[2]
    imgD /= (cube[2] / 2.)
    return imgD, curLabel, np.asarray(cube), com, np.array(M, dtype='float32'), rot
