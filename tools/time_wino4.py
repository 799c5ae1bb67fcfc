#!/usr/bin/env python
"""Times the F(4x4,3x3) conv (and its IN-fused form) at N = 128 / 256 for the library named by LSPS_HIP_LIB."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from lsps_amd import _lib, ops  # noqa: E402
from check_wino4 import conv_in, set_mode, t_ms  # noqa: E402

dev = torch.device('cuda')
torch.manual_seed(0)
set_mode(2)
out = os.path.basename(os.environ.get('LSPS_HIP_LIB', 'default'))
for N in [int(a) for a in (sys.argv[1:] or ['256'])]:
    x = torch.randn(N, 256, 32, 32, device=dev)
    w = torch.randn(256, 256, 3, 3, device=dev) * 0.02
    fl = 2.0 * N * 1024 * 256 * 256 * 9
    ms = t_ms(lambda: ops.conv2d(x, w, None, 1, 1), 20)
    ms2 = t_ms(lambda: conv_in(x, w, None, 0.01), 20)
    res = torch.randn(N, 256, 32, 32, device=dev)
    ms3 = t_ms(lambda: conv_in(x, w, res, -1.0), 20)
    out += '  N=%d: %.3f ms %.0f TF (fused IN + LeakyReLU %.3f ms, fused IN + residual %.3f ms)' % (N, ms, fl / ms / 1e9, ms2, ms3)
print(out)
