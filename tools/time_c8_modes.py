#!/usr/bin/env python
"""Median-of-many timing of the four c8_conv3x3_kernel modes (3x3 256 -> 256 @32x32, N images) for same-box A/B runs of
ablation builds (LSPS_HIP_LIB)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from lsps_amd import _lib  # noqa: E402

L = _lib.lib()
dev = torch.device('cuda')
st = _lib.stream()
BF = torch.bfloat16
N = int(os.environ.get('N', '512'))
reps = int(os.environ.get('REPS', '40'))
C = K = 256
x = torch.randn(N, C // 8, 32, 32, 8, device=dev).to(BF)
r = torch.randn(N, C // 8, 32, 32, 8, device=dev).to(BF)
w = torch.randn(K, C, 3, 3, device=dev) * 0.02
y = torch.empty_like(x)
rstd = torch.empty(N * K, device=dev)
ws, wsb = _lib.workspace(L.lsps_c8_conv3x3_workspace_bytes(C, K), dev)
P = lambda t: t.data_ptr()   # noqa: E731
calls = {
    'conv+IN+LReLU': lambda: L.lsps_c8_conv3x3_in_fwd(P(x), P(w), None, P(y), P(rstd), N, C, 32, 32, K, 0.01, 1e-5, ws, wsb, st),
    'conv+IN+res': lambda: L.lsps_c8_conv3x3_in_fwd(P(x), P(w), P(r), P(y), P(rstd), N, C, 32, 32, K, -1.0, 1e-5, ws, wsb, st),
    'dgrad+INbwd': lambda: L.lsps_c8_conv3x3_dgrad_inbwd(P(r), P(w), P(x), P(rstd), P(y), N, C, 32, 32, K, 0.01, ws, wsb, st),
    'dgrad+skip': lambda: L.lsps_c8_conv3x3_dgrad_acc(P(r), P(w), P(x), P(y), N, C, 32, 32, K, ws, wsb, st),
}
out = []
for name, fn in calls.items():
    ts = []
    for i in range(reps + 5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        _lib.check(fn(), name)
        e1.record()
        e1.synchronize()
        if i >= 5:
            ts.append(e0.elapsed_time(e1))
    ts.sort()
    out.append('%s %.3f' % (name, ts[len(ts) // 2]))
print('N=%d median ms: ' % N + ' | '.join(out))
