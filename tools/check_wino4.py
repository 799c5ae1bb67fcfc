#!/usr/bin/env python
"""Winograd F(4x4,3x3) kernel (conv_wino4.h) vs a float64 convolution: forward, dgrad, the InstanceNorm-fused entry; speed
against the F(2x2,3x3) kernel and the direct one (modes of lsps_set_winograd: 2 = F4 where eligible, 4 = F2 only, 0 = direct)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import torch.nn.functional as F  # noqa: E402
from lsps_amd import _lib, ops  # noqa: E402


def t_ms(fn, iters=10):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


def set_mode(code):
    _lib.check(_lib.lib().lsps_set_winograd(code), 'set_winograd')


def conv_in(x, w, residual, slope):
    L = _lib.lib()
    N, C, H, W = x.shape
    K = w.shape[0]
    y = torch.empty((N, K, H, W), device=x.device)
    rstd = torch.empty(N * K, device=x.device)
    ws, wsb = _lib.workspace(L.lsps_conv2d_workspace_bytes(N, C, H, W, K, 3, 3, 1, 1), x.device)
    _lib.check(L.lsps_conv2d_in_fwd(_lib.ptr(x), _lib.ptr(w), _lib.ptr(residual), _lib.ptr(y), _lib.ptr(rstd), N, C, H, W, K,
                                    slope, 1e-5, ws, wsb, _lib.stream()), 'conv2d_in_fwd')
    return y, rstd


def main():
    dev = torch.device('cuda')
    torch.manual_seed(0)
    for N, C, K in [(2, 256, 256), (3, 64, 96), (1, 8, 32), (5, 128, 64)]:
        x = torch.randn(N, C, 32, 32, device=dev)
        w = torch.randn(K, C, 3, 3, device=dev) * 0.02
        b = torch.randn(K, device=dev)
        ref = F.conv2d(x.double(), w.double(), b.double(), padding=1)
        for mode in (2, 4, 0):
            set_mode(mode)
            ops.kernel_log_begin()
            y = ops.conv2d(x, w, b, 1, 1)
            names = ops.kernel_log_end()
            err = ((y.double() - ref).abs().max() / ref.abs().max()).item()
            print('N=%d C=%d K=%d mode %d %-22s fwd rel err %.2e' % (N, C, K, mode, names[-1], err))
        set_mode(2)
        xg = x.clone().requires_grad_(True)
        ops.kernel_log_begin()
        yy = ops.conv2d(xg, w, b, 1, 1)
        g = torch.randn_like(yy)
        yy.backward(g)
        names = ops.kernel_log_end()
        xr = x.double().clone().requires_grad_(True)
        F.conv2d(xr, w.double(), b.double(), padding=1).backward(g.double())
        print('   dgrad (%s) rel err %.2e' % (names[1] if len(names) > 1 else names, ((xg.grad.double() - xr.grad).abs().max() / xr.grad.abs().max()).item()))
        # fused conv + InstanceNorm (+ LeakyReLU | + residual)
        c0 = F.conv2d(x.double(), w.double(), None, padding=1)
        for residual, slope in ((None, 0.01), (torch.randn(N, K, 32, 32, device=dev), -1.0)):
            y, rstd = conv_in(x, w, residual, slope)
            r = F.instance_norm(c0, eps=1e-5)
            r = F.leaky_relu(r, slope) if residual is None else r + residual.double()
            rr = 1.0 / torch.sqrt(c0.var(dim=(2, 3), unbiased=False) + 1e-5)
            print('   conv+IN%s rel err %.2e   rstd rel err %.2e' % ('+lrelu' if residual is None else '+res',
                  ((y.double() - r).abs().max() / r.abs().max()).item(),
                  ((rstd.double().view(N, K) - rr).abs().max() / rr.abs().max()).item()))
    for N in (16, 32, 128, 256):
        C = K = 256
        x = torch.randn(N, C, 32, 32, device=dev)
        w = torch.randn(K, C, 3, 3, device=dev) * 0.02
        fl = 2.0 * N * 1024 * C * K * 9
        line = 'N=%d' % N
        for mode in (2, 4, 0):
            set_mode(mode)
            ms = t_ms(lambda: ops.conv2d(x, w, None, 1, 1))
            line += '   mode %d: %.3f ms %.0f TF' % (mode, ms, fl / ms / 1e9)
        set_mode(2)
        ms = t_ms(lambda: conv_in(x, w, None, 0.01))
        line += '   F4+IN fused: %.3f ms' % ms
        set_mode(4)
        ms = t_ms(lambda: conv_in(x, w, None, 0.01))
        line += '   F2 + IN pass: %.3f ms' % ms
        print(line)
    set_mode(1)


if __name__ == '__main__':
    main()
