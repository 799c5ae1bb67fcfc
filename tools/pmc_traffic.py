#!/usr/bin/env python
"""Workload for the HBM-traffic PMC passes: a calibration kernel with a known byte count (lsps_axpy on
3 x 1 GiB arrays: 2 GiB read, 1 GiB written, 16 B/lane accesses) followed by the dominant conv layer
(3x3 256->256 @32x32, N=256: forward, dgrad, wgrad).  Run under
  rocprofv3 --kernel-trace --pmc FETCH_SIZE ...   and, separately,   --pmc WRITE_SIZE ..."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from lsps_amd import _lib  # noqa: E402

L = _lib.lib()
dev = torch.device('cuda')
st = _lib.stream()
n = 1 << 28
a, b, c = (torch.ones(n, device=dev) for _ in range(3))
for _ in range(2):
    _lib.check(L.lsps_axpy(a.data_ptr(), b.data_ptr(), 0.5, c.data_ptr(), n, st), 'axpy')
torch.cuda.synchronize()
del a, b, c
N, C, H, K = 256, 256, 32, 256
x = torch.randn(N, C, H, H, device=dev)
w = torch.randn(K, C, 3, 3, device=dev) * 0.05
y = torch.empty(N, K, H, H, device=dev)
dy = torch.randn(N, K, H, H, device=dev)
dx = torch.empty_like(x)
dw = torch.empty_like(w)
ws, wsb = _lib.workspace(L.lsps_conv2d_workspace_bytes(N, C, H, H, K, 3, 3, 1, 1), dev)
for _ in range(2):
    _lib.check(L.lsps_conv2d_fwd(x.data_ptr(), w.data_ptr(), None, y.data_ptr(), N, C, H, H, K, 3, 3, 1, 1, 0, 0.01, ws, wsb, st), 'f')
    _lib.check(L.lsps_conv2d_dgrad(dy.data_ptr(), w.data_ptr(), dx.data_ptr(), N, C, H, H, K, 3, 3, 1, 1, ws, wsb, st), 'd')
    _lib.check(L.lsps_conv2d_wgrad(x.data_ptr(), dy.data_ptr(), dw.data_ptr(), None, N, C, H, H, K, 3, 3, 1, 1, ws, wsb, st), 'w')
torch.cuda.synchronize()
