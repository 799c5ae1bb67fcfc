#!/usr/bin/env python
"""One round of F(4x4,3x3) workgroups on 256 / 128 / 64 / 32 CUs (N = 32 / 16 / 8 / 4 images x 8 k slices, split path off): does a workgroup
run faster when fewer CUs start, load and store at the same moment?  If yes, the lockstep rounds of a full launch pay for their
synchronised prologue loads / epilogue stores."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import torch  # noqa: E402
from lsps_amd import options  # noqa: E402
from check_wino4 import conv_in, set_mode, t_ms  # noqa: E402

dev = torch.device('cuda')
torch.manual_seed(0)
set_mode(2)
options.set(wino4_split=False)
for C in (256, 64):
    row = []
    for N in (4, 8, 16, 32, 64, 128):
        x = torch.randn(N, C, 32, 32, device=dev)
        w = torch.randn(256, C, 3, 3, device=dev) * 0.02
        res = torch.randn(N, 256, 32, 32, device=dev)
        ms = t_ms(lambda: conv_in(x, w, None, 0.01), 30)
        ms2 = t_ms(lambda: conv_in(x, w, res, -1.0), 30)
        row.append("N=%d (%d WGs) %.1f / %.1f us" % (N, N * 8, 1e3 * ms, 1e3 * ms2))
    print("C=%d  IN+LeakyReLU / IN+residual: " % C + "  ".join(row))
