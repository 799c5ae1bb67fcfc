#!/usr/bin/env python
"""Discriminator-trunk layers (3x3 / stride 2 on 32x32 ... 4x4 inputs) in batch-innermost layout (csrc/chwn.hip) against the
NCHW kernels: forward, dgrad, wgrad per layer (HIP events), and the two layout transposes.
usage: python tools/bench_chwn.py [N ...]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from lsps_amd import _lib, ops  # noqa: E402
from check_wino4 import t_ms  # noqa: E402

L = _lib.lib()
dev = torch.device('cuda')
LAYERS = [('dis_s0', 128, 32, 256), ('dis_s1', 256, 16, 512), ('dis_s2', 512, 8, 1024), ('dis_s3', 1024, 4, 2048)]
for N in [int(a) for a in (sys.argv[1:] or ['768', '128'])]:
    tot = dict(chwn=0.0, nchw=0.0)
    for name, C, H, K in LAYERS:
        P = H // 2
        fl = 2.0 * N * K * P * P * C * 9
        x = torch.randn(C, H, H, N, device=dev)
        w = torch.randn(K, C, 3, 3, device=dev) * 0.02
        b = torch.randn(K, device=dev)
        y = torch.empty(K, P, P, N, device=dev)
        dy = torch.randn(K, P, P, N, device=dev)
        dx = torch.empty_like(x)
        dw = torch.empty_like(w)
        ws, wsb = _lib.workspace(L.lsps_conv3x3s2_chwn_workspace_bytes(N, C, H, H, K), dev)
        st = _lib.stream()
        f = t_ms(lambda: L.lsps_conv3x3s2_chwn_fwd(x.data_ptr(), w.data_ptr(), b.data_ptr(), y.data_ptr(), N, C, H, H, K, 1, 0.01, ws, wsb, st))
        d = t_ms(lambda: L.lsps_conv3x3s2_chwn_dgrad(dy.data_ptr(), w.data_ptr(), dx.data_ptr(), N, C, H, H, K, ws, wsb, st))
        g = t_ms(lambda: L.lsps_conv3x3s2_chwn_wgrad(x.data_ptr(), dy.data_ptr(), dw.data_ptr(), N, C, H, H, K, ws, wsb, st))
        xn = torch.randn(N, C, H, H, device=dev)
        yn = torch.empty(N, K, P, P, device=dev)
        dyn = torch.randn(N, K, P, P, device=dev)
        dxn = torch.empty_like(xn)
        ws2, wsb2 = _lib.workspace(L.lsps_conv2d_workspace_bytes(N, C, H, H, K, 3, 3, 2, 1), dev)
        f2 = t_ms(lambda: L.lsps_conv2d_fwd(xn.data_ptr(), w.data_ptr(), b.data_ptr(), yn.data_ptr(), N, C, H, H, K, 3, 3, 2, 1, 1, 0.01, ws2, wsb2, st))
        d2 = t_ms(lambda: L.lsps_conv2d_dgrad(dyn.data_ptr(), w.data_ptr(), dxn.data_ptr(), N, C, H, H, K, 3, 3, 2, 1, ws2, wsb2, st))
        g2 = t_ms(lambda: L.lsps_conv2d_wgrad(xn.data_ptr(), dyn.data_ptr(), dw.data_ptr(), None, N, C, H, H, K, 3, 3, 2, 1, ws2, wsb2, st))
        tot['chwn'] += f + d + g
        tot['nchw'] += f2 + d2 + g2
        print('N=%d %s  chwn fwd %.3f ms %5.1f TF  dgrad %.3f %5.1f  wgrad %.3f %5.1f   |  nchw fwd %.3f %5.1f  dgrad %.3f %5.1f  wgrad %.3f %5.1f'
              % (N, name, f, fl / f / 1e9, d, fl / d / 1e9, g, fl / g / 1e9, f2, fl / f2 / 1e9, d2, fl / d2 / 1e9, g2, fl / g2 / 1e9))
    a = torch.randn(N, 128, 32, 32, device=dev)
    tt = t_ms(lambda: ops.nchw_to_chwn(a))
    c = torch.randn(2048, 2, 2, N, device=dev)
    tb = t_ms(lambda: ops.chwn_to_nchw(c))
    print('N=%d trunk total: chwn %.3f ms (+ transposes in %.3f, out %.3f)   nchw %.3f ms' % (N, tot['chwn'], tt, tb, tot['nchw']))
