#!/usr/bin/env python
"""Direct vs Winograd 3x3 kernels over batch sizes (C-ABI calls, HIP-event timing): where the 'auto' threshold belongs."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from lsps_amd import _lib, ops  # noqa: E402


def t_ms(fn, iters=20):
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
    L = _lib.lib()
    st = _lib.stream()
    C = K = 256
    H = 32
    for N in (1, 2, 4, 6, 8, 12, 16, 24, 32, 64):
        x = torch.randn(N, C, H, 32, device=dev)
        w = torch.randn(K, C, 3, 3, device=dev) * 0.02
        y = torch.empty(N, K, H, 32, device=dev)
        dw = torch.empty_like(w)
        ws, wsb = _lib.workspace(L.lsps_conv2d_workspace_bytes(N, C, H, 32, K, 3, 3, 1, 1), dev)
        row = []
        for mode in ('off', 'always'):
            ops.set_winograd(mode)
            ops.weight_cache_begin(dev)         # as inside a trainer step: transformed weights are cached
            f = t_ms(lambda: _lib.check(L.lsps_conv2d_fwd(_lib.ptr(x), _lib.ptr(w), None, _lib.ptr(y), N, C, H, 32, K, 3, 3, 1, 1,
                                                          0, 0.01, ws, wsb, st), 'f'))
            ops.weight_cache_end()
            g = t_ms(lambda: _lib.check(L.lsps_conv2d_wgrad(_lib.ptr(x), _lib.ptr(y), _lib.ptr(dw), None, N, C, H, 32, K, 3, 3,
                                                            1, 1, ws, wsb, st), 'w'))
            row += [f, g]
        print('N=%3d  fwd direct %.3f wino %.3f   wgrad direct %.3f wino %.3f ms' % (N, row[0], row[2], row[1], row[3]))


if __name__ == '__main__':
    main()
