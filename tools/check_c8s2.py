#!/usr/bin/env python
"""Timing of the C8 stride-2 kernels (csrc/c8s2.h) at BASELINE config 5 sizes: every stride-2 layer of the nets, forward
direction / transposed direction / weight gradient, TFLOP/s algorithmic (2 N K P Q C 9)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from lsps_amd import _lib  # noqa: E402

L = _lib.lib()
dev = torch.device('cuda')
st = _lib.stream()
BF = torch.bfloat16
NG = int(os.environ.get('C8S2_NGEN', '512'))
ND = int(os.environ.get('C8S2_NDIS', '1536'))
# (name, N, C, H, K, transposed-conv layer?)
LAYERS = [('gen down1', NG, 64, 128, 128, 0), ('gen down2', NG, 128, 64, 256, 0), ('gen up1 (convT 256->128)', NG, 128, 64, 256, 1),
          ('gen up2 (convT 128->64)', NG, 64, 128, 128, 1), ('dis front2', ND, 64, 64, 128, 0), ('dis trunk1', ND, 128, 32, 256, 0),
          ('dis trunk2', ND, 256, 16, 512, 0), ('dis trunk3', ND, 512, 8, 1024, 0), ('dis trunk4', ND, 1024, 4, 2048, 0)]
reps = int(os.environ.get('C8S2_REPS', '5'))


def timeit(fn):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


for name, N, C, H, K, tr in LAYERS:
    P = H // 2
    big = torch.randn(N, C // 8, H, H, 8, device=dev).to(BF)
    small = torch.randn(N, K // 8, P, P, 8, device=dev).to(BF)
    bias = torch.randn(K if not tr else C, device=dev)
    ws, wsb = _lib.workspace(L.lsps_c8_conv3x3s2_workspace_bytes(N, C, H, H, K), dev)
    fl = 2.0 * N * K * P * P * C * 9
    if not tr:
        w = torch.randn(K, C, 3, 3, device=dev) * 0.05
        dw = torch.empty_like(w)
        out_s, out_b = torch.empty_like(small), torch.empty_like(big)
        t_f = timeit(lambda: _lib.check(L.lsps_c8_conv3x3s2_fwd(big.data_ptr(), w.data_ptr(), bias.data_ptr(), out_s.data_ptr(), N, C, H, H, K, 0.01, ws, wsb, st), 'f'))
        t_t = timeit(lambda: _lib.check(L.lsps_c8_conv3x3s2_dgrad(small.data_ptr(), w.data_ptr(), out_b.data_ptr(), N, C, H, H, K, ws, wsb, st), 'd'))
        t_w = timeit(lambda: _lib.check(L.lsps_c8_conv3x3s2_wgrad(big.data_ptr(), small.data_ptr(), dw.data_ptr(), N, C, H, H, K, ws, wsb, st), 'w'))
    else:                       # ConvTranspose2d(Ci = K, Co = C): x small -> y big
        w = torch.randn(K, C, 3, 3, device=dev) * 0.05
        dw = torch.empty_like(w)
        out_s, out_b = torch.empty_like(small), torch.empty_like(big)
        t_t = timeit(lambda: _lib.check(L.lsps_c8_convT3x3s2_fwd(small.data_ptr(), w.data_ptr(), bias.data_ptr(), out_b.data_ptr(), N, K, P, P, C, 0.01, ws, wsb, st), 'f'))
        t_f = timeit(lambda: _lib.check(L.lsps_c8_convT3x3s2_dgrad(big.data_ptr(), w.data_ptr(), out_s.data_ptr(), N, K, P, P, C, ws, wsb, st), 'd'))
        t_w = timeit(lambda: _lib.check(L.lsps_c8_convT3x3s2_wgrad(small.data_ptr(), big.data_ptr(), dw.data_ptr(), N, K, P, P, C, ws, wsb, st), 'w'))
    print('%-26s N=%4d C=%4d %3dx%-3d K=%4d  %6.1f GFLOP | fwd-dir %.3f ms %5.0f TF | tr-dir %.3f ms %5.0f TF | wgrad %.3f ms %5.0f TF'
          % (name, N, C, H, H, K, fl / 1e9, t_f, fl / t_f / 1e9, t_t, fl / t_t / 1e9, t_w, fl / t_w / 1e9), flush=True)
