#!/usr/bin/env python
"""Do two HIP streams really run side by side on this box?  Two chains of quarter-chip kernels (1024^3 f32 matmuls = 64
workgroups), one per stream, eager and as two hipGraphs, for several candidate second streams (HIP maps streams to hardware
queues round-robin; queues that share a compute pipe are time-sliced by the command processor, not run concurrently).
Prints the time of the pair against one chain alone: 1.0x = perfect overlap, 2.0x = serial."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lsps_amd import ops  # noqa: E402

dev = torch.device('cuda', 0)
n, L = 2, int(os.environ.get('CHAIN', '100'))
# the library's own quarter-chip kernel: F(2x2,3x3) conv 256 -> 256 @32x32 on 2 images = 64 workgroups of 256 threads (~50 us)
ops.set_winograd('always_f2' if os.environ.get('F2', '1') == '1' else 'auto')
xs = [torch.randn(n, 256, 32, 32, device=dev) for _ in range(2)]
ws = [torch.randn(256, 256, 3, 3, device=dev) * 0.02 for _ in range(2)]


def chain(i):
    with torch.no_grad():
        ops.weight_cache_begin(dev)
        for _ in range(L):
            ops.conv2d(xs[i], ws[i], None, 1, 1)
        ops.weight_cache_end()


def timed(fn, k=5):
    fn()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(k):
        t0 = time.perf_counter()
        fn()
        torch.cuda.synchronize()
        best = min(best, time.perf_counter() - t0)
    return 1e3 * best


main = torch.cuda.current_stream()
chain(0); chain(1)                     # library initialisation outside any capture
torch.cuda.synchronize()
print("GPU_MAX_HW_QUEUES =", os.environ.get('GPU_MAX_HW_QUEUES'), " chain =", L, "launches of a 3x3 conv on", n, "images; kernel:", ops._lib.lib().lsps_last_kernel(None))
g0 = torch.cuda.CUDAGraph()
with torch.cuda.graph(g0):
    chain(0)
t_one = timed(g0.replay)
print("one chain as a hipGraph: %.3f ms (%.1f us per kernel)" % (t_one, 1e3 * t_one / L))
cands = [('default-priority stream #%d' % k, torch.cuda.Stream()) for k in range(6)] + \
        [('high-priority stream #%d' % k, torch.cuda.Stream(priority=-1)) for k in range(3)]
for name, s in cands:
    g1 = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g1, stream=s):
        chain(1)

    def pair():
        s.wait_stream(main)
        g0.replay()
        with torch.cuda.stream(s):
            g1.replay()
        main.wait_stream(s)
    t_pair = timed(pair)

    def pair_eager():
        s.wait_stream(main)
        chain(0)
        with torch.cuda.stream(s):
            chain(1)
        main.wait_stream(s)
    t_eager = timed(pair_eager)
    print("%-28s two graphs %.3f ms = %.2fx one chain | eager pair %.3f ms = %.2fx" % (name, t_pair, t_pair / t_one, t_eager, t_eager / t_one))
# one graph with an internal fork
s = cands[0][1]
g2 = torch.cuda.CUDAGraph()
with torch.cuda.graph(g2):
    s.wait_stream(torch.cuda.current_stream())
    chain(0)
    with torch.cuda.stream(s):
        chain(1)
    torch.cuda.current_stream().wait_stream(s)
t_fork = timed(g2.replay)
print("one graph with a two-branch fork: %.3f ms = %.2fx one chain" % (t_fork, t_fork / t_one))
