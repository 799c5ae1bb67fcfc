#!/usr/bin/env python
"""Fixed cost per workgroup of the F(4x4,3x3) kernel: time of the IN-fused conv (K = 256, 32x32 maps) against the number of INPUT channels
C in {64, 128, 192, 256}: t(C) = a + b C per 256-workgroup round; a = prologue + epilogue + launch share, b C = the main loop.  What a
persistent variant that prefetches the next tile's first rows under the epilogue could hide is (a - epilogue)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import torch  # noqa: E402
from check_wino4 import conv_in, set_mode, t_ms  # noqa: E402

dev = torch.device('cuda')
torch.manual_seed(0)
set_mode(2)
for N in (128, 256):
    pts = []
    for C in (64, 128, 192, 256):
        x = torch.randn(N, C, 32, 32, device=dev)
        w = torch.randn(256, C, 3, 3, device=dev) * 0.02
        ms = t_ms(lambda: conv_in(x, w, None, 0.01), 20)
        pts.append((C, ms))
    n = len(pts)
    sx, sy = sum(c for c, _ in pts), sum(t for _, t in pts)
    sxx, sxy = sum(c * c for c, _ in pts), sum(c * t for c, t in pts)
    b = (n * sxy - sx * sy) / (n * sxx - sx * sx)
    a = (sy - b * sx) / n
    rounds = N * 8 / 256.0
    print("N=%d  " % N + "  ".join("C=%d %.3f ms" % p for p in pts) + "  | fit t = %.3f + %.5f C ms: fixed %.1f us per workgroup round (%.1f %% of the C = 256 "
          "launch), main loop %.1f us per round" % (a, b, 1e3 * a / rounds, 100 * a / pts[-1][1], 1e3 * b * 256 / rounds))
