#!/usr/bin/env python
"""Per-layer micro-benchmark of the conv kernels (HIP events on the launch stream).
usage: python tools/bench_conv.py [--n 256] [--iters 5] [--only res]"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from lsps_amd import _lib  # noqa: E402

# name, (C, H, W, K, R, stride, pad), transposed?
LAYERS = [
    ('res3x3', (256, 32, 32, 256, 3, 1, 1)),
    ('down1', (64, 128, 128, 128, 3, 2, 1)),
    ('down2', (128, 64, 64, 256, 3, 2, 1)),
    ('stem7x7', (1, 128, 128, 64, 7, 1, 3)),
    ('dis_s0', (128, 32, 32, 256, 3, 2, 1)),
    ('dis_s1', (256, 16, 16, 512, 3, 2, 1)),
    ('dis_s2', (512, 8, 8, 1024, 3, 2, 1)),
    ('dis_s3', (1024, 4, 4, 2048, 3, 2, 1)),
    ('disstem', (1, 128, 128, 64, 7, 2, 3)),
    ('dis_f1', (64, 64, 64, 128, 3, 2, 1)),
]
# ConvTranspose layers: name, (Ci, H, W, Co, R, stride, pad, outpad)
T_LAYERS = [
    ('up1', (256, 32, 32, 128, 3, 2, 1, 1)),
    ('up2', (128, 64, 64, 64, 3, 2, 1, 1)),
    ('out1x1', (64, 128, 128, 1, 1, 1, 0, 0)),
]


def time_it(fn, iters):
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
    ap = argparse.ArgumentParser()
    ap.add_argument('--n', type=int, default=256)
    ap.add_argument('--iters', type=int, default=5)
    ap.add_argument('--only', default='')
    ap.add_argument('--ops', default='fwd,dgrad,wgrad')
    ap.add_argument('--math', default='f32')
    a = ap.parse_args()
    L = _lib.lib()
    L.lsps_set_math_mode({'f32': 0, 'bf16': 1, 'f32_split': 2}[a.math])
    dev = torch.device('cuda')
    st = _lib.stream()
    for name, (C, H, W, K, R, s, p) in LAYERS:
        if a.only and a.only not in name:
            continue
        N = a.n
        P = (H + 2 * p - R) // s + 1
        x = torch.randn(N, C, H, W, device=dev)
        w = torch.randn(K, C, R, R, device=dev) * 0.05
        b = torch.randn(K, device=dev)
        y = torch.empty(N, K, P, P, device=dev)
        dy = torch.randn(N, K, P, P, device=dev)
        dx = torch.empty_like(x)
        dw = torch.empty_like(w)
        ws, wsb = _lib.workspace(L.lsps_conv2d_workspace_bytes(N, C, H, W, K, R, R, s, p), dev)
        flops = 2.0 * N * K * P * P * C * R * R
        fns = {
            'fwd': lambda: _lib.check(L.lsps_conv2d_fwd(x.data_ptr(), w.data_ptr(), b.data_ptr(), y.data_ptr(), N, C, H, W,
                                                        K, R, R, s, p, 1, 0.01, ws, wsb, st), 'fwd'),
            'dgrad': lambda: _lib.check(L.lsps_conv2d_dgrad(dy.data_ptr(), w.data_ptr(), dx.data_ptr(), N, C, H, W, K, R, R,
                                                            s, p, ws, wsb, st), 'dgrad'),
            'wgrad': lambda: _lib.check(L.lsps_conv2d_wgrad(x.data_ptr(), dy.data_ptr(), dw.data_ptr(), None, N, C, H, W, K,
                                                            R, R, s, p, ws, wsb, st), 'wgrad'),
        }
        for op in a.ops.split(','):
            ms = time_it(fns[op], a.iters)
            print('%-8s %-6s N=%d  %8.3f ms  %7.1f TFLOP/s (%.0f%% of 157.3)' % (name, op, N, ms, flops / ms / 1e9,
                                                                                  100 * flops / ms / 1e9 / 157.3))
    main_t(a)


def main_t(a):
    L = _lib.lib()
    dev = torch.device('cuda')
    st = _lib.stream()
    for name, (Ci, H, W, Co, R, s, p, op) in T_LAYERS:
        if a.only and a.only not in name:
            continue
        N = a.n
        Ho = (H - 1) * s - 2 * p + R + op
        x = torch.randn(N, Ci, H, W, device=dev)
        w = torch.randn(Ci, Co, R, R, device=dev) * 0.05
        b = torch.randn(Co, device=dev)
        y = torch.empty(N, Co, Ho, Ho, device=dev)
        dy = torch.randn(N, Co, Ho, Ho, device=dev)
        dx = torch.empty_like(x)
        dw = torch.empty_like(w)
        db = torch.empty(Co, device=dev)
        ws, wsb = _lib.workspace(L.lsps_convT2d_workspace_bytes(N, Ci, H, W, Co, R, R, s, p, op), dev)
        flops = 2.0 * N * Ci * H * W * Co * R * R
        fns = {
            'fwd': lambda: _lib.check(L.lsps_convT2d_fwd(x.data_ptr(), w.data_ptr(), b.data_ptr(), y.data_ptr(), N, Ci, H, W,
                                                         Co, R, R, s, p, op, 1, 0.01, ws, wsb, st), 'fwd'),
            'dgrad': lambda: _lib.check(L.lsps_convT2d_dgrad(dy.data_ptr(), w.data_ptr(), dx.data_ptr(), N, Ci, H, W, Co, R,
                                                             R, s, p, op, ws, wsb, st), 'dgrad'),
            'wgrad': lambda: _lib.check(L.lsps_convT2d_wgrad(x.data_ptr(), dy.data_ptr(), dw.data_ptr(), db.data_ptr(), N,
                                                             Ci, H, W, Co, R, R, s, p, op, ws, wsb, st), 'wgrad'),
        }
        for opn in a.ops.split(','):
            ms = time_it(fns[opn], a.iters)
            print('T:%-7s %-6s N=%d  %8.3f ms  %7.1f TFLOP/s' % (name, opn, N, ms, flops / ms / 1e9))


if __name__ == '__main__':
    main()
