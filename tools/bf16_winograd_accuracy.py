#!/usr/bin/env python
"""CPU check quoted in DESIGN.md 3.3: error (relative to the output's abs-max, f64 reference) of a 3x3 conv over 256 channels
when (a) the operands of a direct conv, (b) U and V of Winograd F(2x2,3x3), (c) U and V of F(4x4,3x3) are rounded to bf16."""
import numpy as np
import torch
import torch.nn.functional as F


def bf16(a):
    return torch.tensor(a, dtype=torch.float32).to(torch.bfloat16).to(torch.float64).numpy()


C, K, H = 256, 8, 32
x = np.random.RandomState(0).randn(C, H, H)
w = np.random.RandomState(1).randn(K, C, 3, 3) * 0.02
ref = F.conv2d(torch.tensor(x)[None], torch.tensor(w), padding=1)[0].numpy()
xp = np.pad(x, ((0, 0), (1, 1), (1, 1)))
FORMS = {
    4: (np.array([[4, 0, -5, 0, 1, 0], [0, -4, -4, 1, 1, 0], [0, 4, -4, -1, 1, 0], [0, -2, -1, 2, 1, 0], [0, 2, -1, -2, 1, 0],
                  [0, 4, 0, -5, 0, 1]], float),
        np.array([[1 / 4, 0, 0], [-1 / 6, -1 / 6, -1 / 6], [-1 / 6, 1 / 6, -1 / 6], [1 / 24, 1 / 12, 1 / 6], [1 / 24, -1 / 12, 1 / 6],
                  [0, 0, 1]]),
        np.array([[1, 1, 1, 1, 1, 0], [0, 1, -1, 2, -2, 0], [0, 1, 1, 4, 4, 0], [0, 1, -1, 8, -8, 1]], float)),
    2: (np.array([[1, 0, -1, 0], [0, 1, 1, 0], [0, -1, 1, 0], [0, 1, 0, -1]], float),
        np.array([[1, 0, 0], [.5, .5, .5], [.5, -.5, .5], [0, 0, 1]]),
        np.array([[1, 1, 1, 0], [0, 1, -1, -1]], float)),
}


def winograd(m, rnd):
    BT, G, AT = FORMS[m]
    U = rnd(np.einsum('ia,kcab,jb->kcij', G, w, G))
    out = np.zeros((K, H, H))
    for ty in range(H // m):
        for tx in range(H // m):
            d = xp[:, m * ty:m * ty + m + 2, m * tx:m * tx + m + 2]
            V = rnd(np.einsum('ia,cab,jb->cij', BT, d, BT))
            M = np.einsum('kcij,cij->kij', U, V)
            out[:, m * ty:m * ty + m, m * tx:m * tx + m] = np.einsum('ia,kab,jb->kij', AT, M, AT)
    return out


def err(o):
    return np.abs(o - ref).max() / np.abs(ref).max()


f32 = lambda a: a.astype(np.float32).astype(np.float64)  # noqa: E731
print('direct, bf16 operands        %.2e' % err(F.conv2d(torch.tensor(bf16(x))[None], torch.tensor(bf16(w)), padding=1)[0].numpy()))
for m in (2, 4):
    print('F(%dx%d,3x3), f32 U and V     %.2e' % (m, m, err(winograd(m, f32))))
    print('F(%dx%d,3x3), bf16 U and V    %.2e' % (m, m, err(winograd(m, bf16))))
