#!/usr/bin/env python
"""Workload for PMC passes over the three-limb stride-2 kernels (csrc/x3s2.h): generator down 2 (128 -> 256 @64x64 -> 32x32, N = 256)
and discriminator trunk 2 (256 -> 512 @16x16 -> 8x8, N = 768): forward, dgrad (plain and with the fused mask), weight gradient."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from lsps_amd import _lib, ops  # noqa: E402

L = _lib.lib()
dev = torch.device('cuda')
st = _lib.stream()
BF = torch.bfloat16
for (N, C, H, K) in ((256, 128, 64, 256), (768, 256, 16, 512)):
    P = H // 2
    x = torch.randn(N, C, H, H, device=dev)
    w = torch.randn(K, C, 3, 3, device=dev) * 0.05
    b = torch.randn(K, device=dev)
    dy = torch.randn(N, K, P, P, device=dev)
    xl, dyl = ops.x3_split(x), ops.x3_split(dy)
    y = torch.empty(N, K, P, P, device=dev)
    yl = torch.empty(N, 3, K // 8, P, P, 8, dtype=BF, device=dev)
    dx = torch.empty(N, C, H, H, device=dev)
    dxl = torch.empty_like(xl)
    dw = torch.empty_like(w)
    db = torch.empty(C, device=dev)
    ws, wsb = _lib.workspace(L.lsps_x3_conv3x3s2_workspace_bytes(N, C, H, H, K), dev)
    for _ in range(2):
        _lib.check(L.lsps_x3_conv3x3s2_fwd(_lib.ptr(xl, BF), _lib.ptr(w), _lib.ptr(b), _lib.ptr(y), None, N, C, H, H, K, 0.01, ws, wsb, st), 'f')
        _lib.check(L.lsps_x3_conv3x3s2_fwd(_lib.ptr(xl, BF), _lib.ptr(w), _lib.ptr(b), None, _lib.ptr(yl, BF), N, C, H, H, K, 0.01, ws, wsb, st), 'f3')
        _lib.check(L.lsps_x3_conv3x3s2_dgrad(_lib.ptr(dyl, BF), _lib.ptr(w), _lib.ptr(dx), None, None, 0.0, None, N, C, H, H, K, ws, wsb, st), 'd')
        _lib.check(L.lsps_x3_conv3x3s2_dgrad(_lib.ptr(dyl, BF), _lib.ptr(w), None, _lib.ptr(dxl, BF), _lib.ptr(xl, BF), 0.01, _lib.ptr(db), N, C, H, H, K,
                                             ws, wsb, st), 'dm')
        _lib.check(L.lsps_x3_conv3x3s2_wgrad(_lib.ptr(xl, BF), _lib.ptr(dyl, BF), _lib.ptr(dw), N, C, H, H, K, ws, wsb, st), 'w')
    torch.cuda.synchronize()
