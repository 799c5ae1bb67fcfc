#!/usr/bin/env python
"""Workload for PMC passes over the stride-2 3x3 kernels: down1 (64 -> 128 @128x128 -> 64x64) forward, dgrad, wgrad at N=256."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from lsps_amd import _lib  # noqa: E402

L = _lib.lib()
dev = torch.device('cuda')
st = _lib.stream()
N, C, H, K = 256, 64, 128, 128
x = torch.randn(N, C, H, H, device=dev)
w = torch.randn(K, C, 3, 3, device=dev) * 0.05
b = torch.randn(K, device=dev)
y = torch.empty(N, K, H // 2, H // 2, device=dev)
dy = torch.randn_like(y)
dx = torch.empty_like(x)
dw = torch.empty_like(w)
ws, wsb = _lib.workspace(L.lsps_conv2d_workspace_bytes(N, C, H, H, K, 3, 3, 2, 1), dev)
for _ in range(3):
    _lib.check(L.lsps_conv2d_fwd(x.data_ptr(), w.data_ptr(), b.data_ptr(), y.data_ptr(), N, C, H, H, K, 3, 3, 2, 1, 1, 0.01, ws, wsb, st), 'f')
    _lib.check(L.lsps_conv2d_dgrad(dy.data_ptr(), w.data_ptr(), dx.data_ptr(), N, C, H, H, K, 3, 3, 2, 1, ws, wsb, st), 'd')
    _lib.check(L.lsps_conv2d_wgrad(x.data_ptr(), dy.data_ptr(), dw.data_ptr(), None, N, C, H, H, K, 3, 3, 2, 1, ws, wsb, st), 'w')
torch.cuda.synchronize()
