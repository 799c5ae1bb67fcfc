"""The algebra the Winograd kernels (lsps_amd/csrc/conv_wino.h) rely on, restated in numpy and checked against a direct
3x3 correlation on CPU: the forward identity, the weight-gradient identity, the position-half split of the forward
kernel (two waves each holding two rows of the 4x4 grid) and the sign convention of the weight-gradient kernel
(A's last row/column taken as +1, undone by G' = diag(1,1,1,-1) G in the reduce kernel)."""
import numpy as np

G = np.array([[1, 0, 0], [.5, .5, .5], [.5, -.5, .5], [0, 0, 1]], dtype=np.float64)
BT = np.array([[1, 0, -1, 0], [0, 1, 1, 0], [0, -1, 1, 0], [0, 1, 0, -1]], dtype=np.float64)
AT = np.array([[1, 1, 1, 0], [0, 1, -1, -1]], dtype=np.float64)


def direct_tile(d, g):
    """2x2 outputs of the 3x3 correlation on a 4x4 input tile."""
    return np.array([[(d[i:i + 3, j:j + 3] * g).sum() for j in range(2)] for i in range(2)])


def test_forward_identity_and_position_halves():
    rng = np.random.default_rng(0)
    for _ in range(20):
        d, g = rng.standard_normal((4, 4)), rng.standard_normal((3, 3))
        U, V = G @ g @ G.T, BT @ d @ BT.T
        M = U * V
        assert np.allclose(AT @ M @ AT.T, direct_tile(d, g), atol=1e-12)
        # kernel epilogue: rows (x, y) = (M[2wp], M[2wp+1]); wp = 0 contributes (x + y, y), wp = 1 (x, -(x + y)),
        # then the column stage (m0 + m1 + m2, m1 - m2 - m3) on both and the halves are added
        total = np.zeros((2, 2))
        for wp in range(2):
            x, y = M[2 * wp], M[2 * wp + 1]
            p0, p1 = (x, -(x + y)) if wp else (x + y, y)
            for a, pr in enumerate((p0, p1)):
                total[a] += [pr[0] + pr[1] + pr[2], pr[1] - pr[2] - pr[3]]
        assert np.allclose(total, direct_tile(d, g), atol=1e-12)


def test_kernel_row_order_of_the_data_transform():
    """wp = 0 reads d rows (0, 2, 1), wp = 1 reads (2, 1, 3); slot 0 = e0 - e1, slot 1 = e1 + sgn * e2."""
    rng = np.random.default_rng(1)
    d = rng.standard_normal((4, 4))
    t = BT @ d
    for wp, rows, sgn in ((0, (0, 2, 1), 1.0), (1, (2, 1, 3), -1.0)):
        e0, e1, e2 = (d[r] for r in rows)
        assert np.allclose(e0 - e1, t[2 * wp]) and np.allclose(e1 + sgn * e2, t[2 * wp + 1])
    # column stage on pairs: (v0, v1) = (t0 - t2, t1 + t2), (v2, v3) = (t2 - t1, t1 - t3)
    V = t @ BT.T
    for i in range(4):
        assert np.allclose([t[i, 0] - t[i, 2], t[i, 1] + t[i, 2], t[i, 2] - t[i, 1], t[i, 1] - t[i, 3]], V[i])


def test_weight_gradient_identity_with_sign_convention():
    rng = np.random.default_rng(2)
    Gp = np.diag([1, 1, 1, -1.0]) @ G
    Ap = np.array([[1, 0], [1, 1], [1, -1], [0, 1.0]])          # A with its last row taken as (0, +1)
    acc = np.zeros((4, 4))
    want = np.zeros((3, 3))
    for _ in range(7):                                           # sum over tiles
        d, dy = rng.standard_normal((4, 4)), rng.standard_normal((2, 2))
        acc += (Ap @ dy @ Ap.T) * (BT @ d @ BT.T)
        for r in range(3):
            for s in range(3):
                want[r, s] += (dy * d[r:r + 2, s:s + 2]).sum()
    assert np.allclose(Gp.T @ acc @ Gp, want, atol=1e-12)
    # A side as the kernel computes it: slot 0 = D0 + b0 D1, slot 1 = a1 D0 + D1; per row (x, y): x, x + y, x - y, y
    dy = rng.standard_normal((2, 2))
    T = Ap @ dy @ Ap.T
    for wp, b0, a1 in ((0, 0.0, 1.0), (1, -1.0, 0.0)):
        for slot, row in enumerate((dy[0] + b0 * dy[1], a1 * dy[0] + dy[1])):
            x, y = row
            assert np.allclose([x, x + y, x - y, y], T[2 * wp + slot])
