#!/usr/bin/env python
"""Winograd F(4x4,3x3) weight-gradient kernel (conv_wino4w.h) vs a float64 reference, and its speed against the F(2x2,3x3)
kernel and the direct one (modes of lsps_set_winograd: 2 = F4 where eligible, 4 = F2 only, 0 = direct)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import torch.nn.functional as F  # noqa: E402
from lsps_amd import _lib, ops  # noqa: E402
from check_wino4 import set_mode, t_ms  # noqa: E402


def wgrad(x, dy):
    L = _lib.lib()
    N, C, H, W = x.shape
    K = dy.shape[1]
    dw = torch.empty((K, C, 3, 3), device=x.device)
    ws, wsb = _lib.workspace(L.lsps_conv2d_workspace_bytes(N, C, H, W, K, 3, 3, 1, 1), x.device)
    _lib.check(L.lsps_conv2d_wgrad(_lib.ptr(x), _lib.ptr(dy), _lib.ptr(dw), None, N, C, H, W, K, 3, 3, 1, 1, ws, wsb,
                                   _lib.stream()), 'conv2d_wgrad')
    return dw


def main():
    dev = torch.device('cuda')
    torch.manual_seed(0)
    for N, C, K, H in [(2, 256, 256, 32), (3, 64, 128, 32), (1, 32, 64, 32), (5, 96, 64, 32), (7, 64, 64, 8), (16, 32, 64, 4), (17, 256, 256, 32)]:
        x = torch.randn(N, C, H, 32, device=dev)
        dy = torch.randn(N, K, H, 32, device=dev)
        w = torch.zeros(K, C, 3, 3, device=dev, dtype=torch.float64, requires_grad=True)
        F.conv2d(x.double(), w, None, padding=1).backward(dy.double())
        ref = w.grad
        for mode in (2, 4, 0):
            set_mode(mode)
            ops.kernel_log_begin()
            dw = wgrad(x, dy)
            names = ops.kernel_log_end()
            err = ((dw.double() - ref).abs().max() / ref.abs().max()).item()
            print('N=%d C=%d K=%d H=%d mode %d %-22s wgrad rel err %.2e' % (N, C, K, H, mode, names[-1] if names else '?', err))
    for N in (8, 16, 32, 64, 128, 256):
        C = K = 256
        x = torch.randn(N, C, 32, 32, device=dev)
        dy = torch.randn(N, K, 32, 32, device=dev)
        fl = 2.0 * N * 1024 * C * K * 9
        line = 'N=%d' % N
        for mode in (2, 4, 0):
            set_mode(mode)
            ms = t_ms(lambda: wgrad(x, dy))
            line += '   mode %d: %.3f ms %.0f TF' % (mode, ms, fl / ms / 1e9)
        print(line)
    set_mode(1)


if __name__ == '__main__':
    main()
