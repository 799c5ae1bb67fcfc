#!/usr/bin/env python
"""How much would running a residual conv's weight gradient BESIDE the input-gradient chain buy?  The two are independent in backward
(the wgrad only feeds the gradient arena); each kernel fills the chip by itself (one workgroup per CU), so a second stream can only
fill launch gaps and tails.  Measures 3x3 256->256 @32x32: K x [dgrad (F(4x4,3x3) with the fused norm backward), wgrad] on ONE stream
against the dgrad chain on stream A and the wgrad chain on stream B.  Usage: python tools/two_stream_wino.py [N]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from lsps_amd import _lib  # noqa: E402

L = _lib.lib()
dev = torch.device('cuda')
N = int(sys.argv[1]) if len(sys.argv) > 1 else 128
C = K = 256
H = 32
REP = 20
x = torch.randn(N, C, H, H, device=dev)
w = torch.randn(K, C, 3, 3, device=dev) * 0.05
dy = torch.randn(N, K, H, H, device=dev)
dx = torch.empty_like(x)
dw = torch.empty_like(w)
sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
wsb = L.lsps_conv2d_workspace_bytes(N, C, H, H, K, 3, 3, 1, 1)
ws = [torch.empty(wsb, dtype=torch.uint8, device=dev) for _ in range(2)]


def dgrad(st, k):
    _lib.check(L.lsps_conv2d_dgrad(dy.data_ptr(), w.data_ptr(), dx.data_ptr(), N, C, H, H, K, 3, 3, 1, 1, ws[k].data_ptr(), wsb, st), 'd')


def wgrad(st, k):
    _lib.check(L.lsps_conv2d_wgrad(x.data_ptr(), dy.data_ptr(), dw.data_ptr(), None, N, C, H, H, K, 3, 3, 1, 1, ws[k].data_ptr(), wsb, st), 'w')


def timed(fn):
    fn()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        e0.record()
        fn()
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1))
    return best


def one_stream():
    st = torch.cuda.current_stream().cuda_stream
    for _ in range(REP):
        dgrad(st, 0)
        wgrad(st, 0)


def only(which):
    def f():
        st = torch.cuda.current_stream().cuda_stream
        for _ in range(REP):
            (dgrad if which == 'd' else wgrad)(st, 0)
    return f


def two_streams():
    cur = torch.cuda.current_stream()
    ev = torch.cuda.Event()
    ev.record(cur)
    sa.wait_event(ev)
    sb.wait_event(ev)
    for _ in range(REP):
        dgrad(sa.cuda_stream, 0)
        wgrad(sb.cuda_stream, 1)
    ea, eb = torch.cuda.Event(), torch.cuda.Event()
    ea.record(sa)
    eb.record(sb)
    cur.wait_event(ea)
    cur.wait_event(eb)


t1, td, tw, t2 = timed(one_stream), timed(only('d')), timed(only('w')), timed(two_streams)
print("N=%d  %d x [dgrad, wgrad]: one stream %.3f ms (dgrad alone %.3f + wgrad alone %.3f = %.3f) | two streams %.3f ms = %.3f of one stream"
      % (N, REP, t1, td, tw, td + tw, t2, t2 / t1))
