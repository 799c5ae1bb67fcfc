#!/usr/bin/env python
"""Winograd 3x3 kernel vs the direct kernel and a float64 reference: error and speed (run with LSPS_WINO=1 / 0)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import torch.nn.functional as F  # noqa: E402
from lsps_amd import ops  # noqa: E402


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


def main():
    dev = torch.device('cuda')
    torch.manual_seed(0)
    ops.set_winograd(os.environ.get('LSPS_WINO_MODE', 'always'))
    for N, C, K, H in [(8, 256, 256, 32), (3, 64, 128, 16), (2, 64, 64, 4), (128, 256, 256, 32), (256, 256, 256, 32)]:
        x = torch.randn(N, C, H, 32, device=dev)
        w = torch.randn(K, C, 3, 3, device=dev) * 0.02
        b = torch.randn(K, device=dev)
        y = ops.conv2d(x, w, b, 1, 1)
        if N <= 8:
            ref = F.conv2d(x.double(), w.double(), b.double(), padding=1)
            err = ((y.double() - ref).abs().max() / ref.abs().max()).item()
        else:
            err = float('nan')
        ms = t_ms(lambda: ops.conv2d(x, w, b, 1, 1))
        fl = 2.0 * N * H * 32 * C * K * 9
        print('N=%d C=%d K=%d H=%d  rel err %.2e  %.3f ms  %.1f TFLOP/s (direct-equivalent)' % (N, C, K, H, err, ms, fl / ms / 1e9))
        # weight gradient (C-ABI call: no bias gradient pass in the timing)
        from lsps_amd import _lib
        L = _lib.lib()
        g = torch.randn(N, K, H, 32, device=dev)
        dw = torch.empty_like(w)
        ws, wsb = _lib.workspace(L.lsps_conv2d_workspace_bytes(N, C, H, 32, K, 3, 3, 1, 1), dev)

        def wg():
            _lib.check(L.lsps_conv2d_wgrad(_lib.ptr(x), _lib.ptr(g), _lib.ptr(dw), None, N, C, H, 32, K, 3, 3, 1, 1, ws, wsb,
                                           _lib.stream()), 'wgrad')
        ms = t_ms(wg)
        werr = float('nan')
        if N <= 8:
            wr = w.double().clone().requires_grad_(True)
            F.conv2d(x.double(), wr, None, padding=1).backward(g.double())
            werr = ((dw.double() - wr.grad).abs().max() / wr.grad.abs().max()).item()
        print('   wgrad rel err %.2e  %.3f ms  %.1f TFLOP/s' % (werr, ms, fl / ms / 1e9))
        # dgrad through autograd
        if N <= 8:
            xg = x.clone().requires_grad_(True)
            yy = ops.conv2d(xg, w, b, 1, 1)
            g = torch.randn_like(yy)
            yy.backward(g)
            xr = x.double().clone().requires_grad_(True)
            F.conv2d(xr, w.double(), b.double(), padding=1).backward(g.double())
            print('   dgrad rel err %.2e' % ((xg.grad.double() - xr.grad).abs().max() / xr.grad.abs().max()).item())


if __name__ == '__main__':
    main()
