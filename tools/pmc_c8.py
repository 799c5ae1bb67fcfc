#!/usr/bin/env python
"""Workload for PMC / timing passes over the bf16 C8 kernels (csrc/c8conv.h, c8wgrad.h, c8s2.h) at BASELINE config 5's
launch shape: 3x3 256 -> 256 @32x32 on N = 512 images (256 per domain): conv + InstanceNorm + LeakyReLU, conv + InstanceNorm +
residual, dgrad through the norm backward, dgrad + skip, weight gradient; plus one stride-2 layer (generator down 2) in the
three directions and the calibration kernel with a known byte count (lsps_axpy, 2 GiB read / 1 GiB written)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from lsps_amd import _lib  # noqa: E402

L = _lib.lib()
dev = torch.device('cuda')
st = _lib.stream()
BF = torch.bfloat16
N = int(os.environ.get('PMC_N', '512'))
reps = int(os.environ.get('PMC_REPS', '3'))
if os.environ.get('PMC_CAL', '1') == '1':
    n = 1 << 28
    a, b, c = (torch.ones(n, device=dev) for _ in range(3))
    for _ in range(2):
        _lib.check(L.lsps_axpy(a.data_ptr(), b.data_ptr(), 0.5, c.data_ptr(), n, st), 'axpy')
    torch.cuda.synchronize()
    del a, b, c
C = K = 256
x = torch.randn(N, C // 8, 32, 32, 8, device=dev).to(BF)
r = torch.randn(N, C // 8, 32, 32, 8, device=dev).to(BF)
w = torch.randn(K, C, 3, 3, device=dev) * 0.02
y = torch.empty_like(x)
y2 = torch.empty_like(x)
dx = torch.empty_like(x)
rstd = torch.empty(N * K, device=dev)
dw = torch.empty_like(w)
ws, wsb = _lib.workspace(max(L.lsps_c8_conv3x3_workspace_bytes(C, K), L.lsps_c8_conv3x3_wgrad_workspace_bytes(N, C, K)), dev)
ev = [torch.cuda.Event(enable_timing=True) for _ in range(6)]
P = lambda t: t.data_ptr()   # noqa: E731
for it in range(reps):
    ev[0].record()
    _lib.check(L.lsps_c8_conv3x3_in_fwd(P(x), P(w), None, P(y), P(rstd), N, C, 32, 32, K, 0.01, 1e-5, ws, wsb, st), 'in1')
    ev[1].record()
    _lib.check(L.lsps_c8_conv3x3_in_fwd(P(y), P(w), P(r), P(y2), P(rstd), N, C, 32, 32, K, -1.0, 1e-5, ws, wsb, st), 'in2')
    ev[2].record()
    _lib.check(L.lsps_c8_conv3x3_dgrad_inbwd(P(r), P(w), P(y), P(rstd), P(dx), N, C, 32, 32, K, 0.01, ws, wsb, st), 'dinb')
    ev[3].record()
    _lib.check(L.lsps_c8_conv3x3_dgrad_acc(P(r), P(w), P(x), P(dx), N, C, 32, 32, K, ws, wsb, st), 'dacc')
    ev[4].record()
    _lib.check(L.lsps_c8_conv3x3_wgrad(P(x), P(r), P(dw), N, C, 32, 32, K, ws, wsb, st), 'wg')
    ev[5].record()
torch.cuda.synchronize()
fl = 2.0 * N * K * 1024 * C * 9
for name, i in (('conv+IN+LReLU', 0), ('conv+IN+res', 1), ('dgrad+INbwd', 2), ('dgrad+skip', 3), ('wgrad', 4)):
    ms = ev[i].elapsed_time(ev[i + 1])
    print('%-14s N=%d  %.3f ms  %.0f TFLOP/s' % (name, N, ms, fl / ms / 1e9))
# stride-2: generator down 2 (128 -> 256, 64x64 -> 32x32)
Cs, Ks, H = 128, 256, 64
big = torch.randn(N, Cs // 8, H, H, 8, device=dev).to(BF)
small = torch.randn(N, Ks // 8, H // 2, H // 2, 8, device=dev).to(BF)
w2 = torch.randn(Ks, Cs, 3, 3, device=dev) * 0.05
bias = torch.randn(Ks, device=dev)
dw2 = torch.empty_like(w2)
o_s, o_b = torch.empty_like(small), torch.empty_like(big)
ws, wsb = _lib.workspace(L.lsps_c8_conv3x3s2_workspace_bytes(N, Cs, H, H, Ks), dev)
for it in range(reps):
    ev[0].record()
    _lib.check(L.lsps_c8_conv3x3s2_fwd(P(big), P(w2), P(bias), P(o_s), N, Cs, H, H, Ks, 0.01, ws, wsb, st), 'f')
    ev[1].record()
    _lib.check(L.lsps_c8_conv3x3s2_dgrad(P(small), P(w2), P(o_b), N, Cs, H, H, Ks, ws, wsb, st), 'd')
    ev[2].record()
    _lib.check(L.lsps_c8_conv3x3s2_wgrad(P(big), P(small), P(dw2), N, Cs, H, H, Ks, ws, wsb, st), 'w')
    ev[3].record()
torch.cuda.synchronize()
fl = 2.0 * N * Ks * (H // 2) ** 2 * Cs * 9
for name, i in (('s2 fwd', 0), ('s2 dgrad', 1), ('s2 wgrad', 2)):
    ms = ev[i].elapsed_time(ev[i + 1])
    print('%-14s N=%d  %.3f ms  %.0f TFLOP/s' % (name, N, ms, fl / ms / 1e9))
