#!/usr/bin/env python
"""Workload for PMC / timing passes over the bf16 residual-conv path (math mode 1): 3x3 256 -> 256 @32x32, N images
(default 512 = BASELINE config 5's 256 per domain): conv + InstanceNorm forward, dgrad through norm backward, wgrad."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from lsps_amd import _lib, ops  # noqa: E402

L = _lib.lib()
dev = torch.device('cuda')
st = _lib.stream()
N = int(os.environ.get('PMC_N', '512'))
C = K = 256
H = 32
ops.set_math_mode('bf16')
x = torch.randn(N, C, H, H, device=dev)
w = torch.randn(K, C, 3, 3, device=dev) * 0.02
y = torch.empty(N, K, H, H, device=dev)
r = torch.empty(N * K, device=dev)
dy = torch.randn_like(y)
dx = torch.empty_like(x)
dw = torch.empty_like(w)
ws, wsb = _lib.workspace(L.lsps_conv2d_workspace_bytes(N, C, H, H, K, 3, 3, 1, 1), dev)
reps = int(os.environ.get('PMC_REPS', '3'))
ev = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
for it in range(reps):
    ev[0].record()
    _lib.check(L.lsps_conv2d_in_fwd(x.data_ptr(), w.data_ptr(), None, y.data_ptr(), r.data_ptr(), N, C, H, H, K, 0.01, 1e-5, ws, wsb, st), 'f')
    ev[1].record()
    _lib.check(L.lsps_conv2d_dgrad_inbwd(dy.data_ptr(), w.data_ptr(), y.data_ptr(), r.data_ptr(), dx.data_ptr(), N, K, H, H, K, 0.01, ws, wsb, st), 'd')
    ev[2].record()
    _lib.check(L.lsps_conv2d_wgrad(x.data_ptr(), dy.data_ptr(), dw.data_ptr(), None, N, C, H, H, K, 3, 3, 1, 1, ws, wsb, st), 'w')
    ev[3].record()
torch.cuda.synchronize()
fl = 2.0 * N * K * H * H * C * 9
for name, a, b in (('conv+IN fwd', 0, 1), ('dgrad+IN bwd', 1, 2), ('wgrad', 2, 3)):
    ms = ev[a].elapsed_time(ev[b])
    print('%-14s N=%d  %.3f ms  %.0f TFLOP/s' % (name, N, ms, fl / ms / 1e9))
