#!/usr/bin/env python
"""Times the F(4x4,3x3) weight-gradient kernel (conv_wino4w.h) at N = 128 / 256 for the library named by LSPS_HIP_LIB."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from check_wino4 import set_mode, t_ms  # noqa: E402
from check_wino4w import wgrad  # noqa: E402

dev = torch.device('cuda')
torch.manual_seed(0)
set_mode(2)
out = os.path.basename(os.environ.get('LSPS_HIP_LIB', 'default'))
for N in [int(a) for a in (sys.argv[1:] or ['128', '256'])]:
    x = torch.randn(N, 256, 32, 32, device=dev)
    dy = torch.randn(N, 256, 32, 32, device=dev)
    fl = 2.0 * N * 1024 * 256 * 256 * 9
    ms = t_ms(lambda: wgrad(x, dy), 20)
    out += '  N=%d: %.3f ms %.0f TF' % (N, ms, fl / ms / 1e9)
print(out)
