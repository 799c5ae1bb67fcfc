#!/usr/bin/env python
"""Round-5 gated experiment: the three-limb ("X3") stride-2 forward conv on the bf16 matrix pipe
(csrc/x3s2.h) against the exact-f32 MFMA kernels it would replace.

1. correctness: error against an f64 convolution (CPU) next to the exact-f32 kernel's error on the same inputs; the split /
   join round trip is exact;
2. timing at the f32 headline's launch sizes: split pass, X3 kernel with f32 NCHW output, X3 kernel with X3 output, and the
   f32 kernel (`lsps_conv2d_fwd` -> igemm_f3x3s2_kernel; trunk layers: `lsps_conv3x3s2_chwn_fwd`)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import torch.nn.functional as F  # noqa: E402  (CPU f64 reference only)
from lsps_amd import _lib  # noqa: E402

L = _lib.lib()
dev = torch.device('cuda')
st = _lib.stream()
reps = int(os.environ.get('X3_REPS', '10'))


def timeit(fn):
    fn()
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


def x3_bytes(N, C, HW):
    return N * 3 * C * HW * 2


def run_x3(x, w, b, K, slope=0.01, out3=False):
    N, C, H, W = x.shape
    xl = torch.empty(x3_bytes(N, C, H * W), dtype=torch.uint8, device=dev)
    _lib.check(L.lsps_x3_split_nchw(x.data_ptr(), xl.data_ptr(), N, C, H * W, st), 'split')
    ws, wsb = _lib.workspace(L.lsps_x3_conv3x3s2_workspace_bytes(N, C, H, W, K), dev)
    P, Q = H // 2, W // 2
    y = torch.empty(N, K, P, Q, device=dev)
    if out3:
        yl = torch.empty(x3_bytes(N, K, P * Q), dtype=torch.uint8, device=dev)
        _lib.check(L.lsps_x3_conv3x3s2_fwd(xl.data_ptr(), w.data_ptr(), b.data_ptr(), None, yl.data_ptr(), N, C, H, W, K, slope, ws, wsb, st), 'x3')
        _lib.check(L.lsps_x3_join_nchw(yl.data_ptr(), y.data_ptr(), N, K, P * Q, st), 'join')
    else:
        _lib.check(L.lsps_x3_conv3x3s2_fwd(xl.data_ptr(), w.data_ptr(), b.data_ptr(), y.data_ptr(), None, N, C, H, W, K, slope, ws, wsb, st), 'x3')
    return y


def run_f32(x, w, b, K, slope=0.01):
    N, C, H, W = x.shape
    ws, wsb = _lib.workspace(L.lsps_conv2d_workspace_bytes(N, C, H, W, K, 3, 3, 2, 1), dev)
    y = torch.empty(N, K, H // 2, W // 2, device=dev)
    _lib.check(L.lsps_conv2d_fwd(x.data_ptr(), w.data_ptr(), b.data_ptr(), y.data_ptr(), N, C, H, W, K, 3, 3, 2, 1, _lib.ACT_LRELU, slope,
                                 ws, wsb, st), 'f32')
    return y


print('== correctness (error relative to the output abs-max, against an f64 CPU convolution)')
torch.manual_seed(0)
worst_ratio = 0.0
for (N, C, H, K) in [(2, 64, 128, 128), (3, 128, 64, 256), (2, 64, 64, 128), (5, 128, 32, 256), (5, 256, 16, 512), (19, 512, 8, 1024)]:
    assert L.lsps_x3_conv3x3s2_ok(N, C, H, H, K), (N, C, H, K)
    x = torch.randn(N, C, H, H, device=dev)
    w = torch.randn(K, C, 3, 3, device=dev) * 0.05
    b = torch.randn(K, device=dev)
    # round trip
    xl = torch.empty(x3_bytes(N, C, H * H), dtype=torch.uint8, device=dev)
    xr = torch.empty_like(x)
    _lib.check(L.lsps_x3_split_nchw(x.data_ptr(), xl.data_ptr(), N, C, H * H, st), 'split')
    _lib.check(L.lsps_x3_join_nchw(xl.data_ptr(), xr.data_ptr(), N, C, H * H, st), 'join')
    assert torch.equal(x, xr), 'split/join is not exact'
    ref = F.leaky_relu(F.conv2d(x.double().cpu(), w.double().cpu(), b.double().cpu(), stride=2, padding=1), 0.01)
    am = float(ref.abs().max())
    e3 = float((run_x3(x, w, b, K).double().cpu() - ref).abs().max()) / am
    e3o = float((run_x3(x, w, b, K, out3=True).double().cpu() - ref).abs().max()) / am
    ef = float((run_f32(x, w, b, K).double().cpu() - ref).abs().max()) / am
    worst_ratio = max(worst_ratio, e3 / ef)
    print('N=%3d C=%4d %3dx%-3d K=%4d  x3 %.2e  x3(X3 out) %.2e  exact-f32 kernel %.2e  ratio %.2f' % (N, C, H, H, K, e3, e3o, ef, e3 / ef),
          flush=True)
print('worst x3 / f32 error ratio: %.2f (gate: <= 1.5)' % worst_ratio)


def x3_split(t):
    N, C, H, W = t.shape
    xl = torch.empty(x3_bytes(N, C, H * W), dtype=torch.uint8, device=dev)
    _lib.check(L.lsps_x3_split_nchw(t.data_ptr(), xl.data_ptr(), N, C, H * W, st), 'split')
    return xl


def x3_join(xl, N, C, H, W):
    y = torch.empty(N, C, H, W, device=dev)
    _lib.check(L.lsps_x3_join_nchw(xl.data_ptr(), y.data_ptr(), N, C, H * W, st), 'join')
    return y


print('== transposed direction (conv dgrad, optionally with the previous layer\'s LeakyReLU backward + bias gradient) and weight gradient')
for (N, C, H, K) in [(2, 64, 128, 128), (3, 128, 64, 256), (5, 128, 32, 256), (5, 256, 16, 512), (19, 512, 8, 1024), (70, 1024, 4, 2048)]:
    if not L.lsps_x3_conv3x3s2_ok(N, C, H, H, K):
        print('N=%d C=%d %dx%d K=%d: geometry not supported' % (N, C, H, H, K))
        continue
    P = H // 2
    x = torch.randn(N, C, H, H, device=dev)
    dy = torch.randn(N, K, P, P, device=dev)
    w = torch.randn(K, C, 3, 3, device=dev) * 0.05
    yprev = torch.randn(N, C, H, H, device=dev)                      # "saved output" of the layer in front
    ws, wsb = _lib.workspace(L.lsps_x3_conv3x3s2_workspace_bytes(N, C, H, H, K), dev)
    dyl, xl, ypl = x3_split(dy), x3_split(x), x3_split(yprev)
    dx_ref = torch.nn.grad.conv2d_input((N, C, H, H), w.double().cpu(), dy.double().cpu(), stride=2, padding=1)
    am = float(dx_ref.abs().max())
    dx = torch.empty(N, C, H, H, device=dev)
    _lib.check(L.lsps_x3_conv3x3s2_dgrad(dyl.data_ptr(), w.data_ptr(), dx.data_ptr(), None, None, 0.0, None, N, C, H, H, K, ws, wsb, st), 'dgrad')
    e_d = float((dx.double().cpu() - dx_ref).abs().max()) / am
    dxl = torch.empty(x3_bytes(N, C, H * H), dtype=torch.uint8, device=dev)
    _lib.check(L.lsps_x3_conv3x3s2_dgrad(dyl.data_ptr(), w.data_ptr(), None, dxl.data_ptr(), None, 0.0, None, N, C, H, H, K, ws, wsb, st), 'dgrad3')
    e_d3 = float((x3_join(dxl, N, C, H, H).double().cpu() - dx_ref).abs().max()) / am
    # masked: g = dx * (yprev > 0 ? 1 : 0.01), db = sum g
    g_ref = torch.where(yprev.double().cpu() > 0, dx_ref, dx_ref * 0.01)
    db = torch.empty(C, device=dev)
    _lib.check(L.lsps_x3_conv3x3s2_dgrad(dyl.data_ptr(), w.data_ptr(), None, dxl.data_ptr(), ypl.data_ptr(), 0.01, db.data_ptr(), N, C, H, H, K, ws, wsb, st), 'dgradm')
    e_m = float((x3_join(dxl, N, C, H, H).double().cpu() - g_ref).abs().max()) / am
    db_ref = g_ref.sum((0, 2, 3))
    e_db = float((db.double().cpu() - db_ref).abs().max() / db_ref.abs().max())
    dxm = torch.empty(N, C, H, H, device=dev)
    _lib.check(L.lsps_x3_conv3x3s2_dgrad(dyl.data_ptr(), w.data_ptr(), dxm.data_ptr(), None, ypl.data_ptr(), 0.01, db.data_ptr(), N, C, H, H, K, ws, wsb, st), 'dgradm32')
    e_m32 = float((dxm.double().cpu() - g_ref).abs().max()) / am
    # f32 kernel for comparison
    wsf, wsfb = _lib.workspace(L.lsps_conv2d_workspace_bytes(N, C, H, H, K, 3, 3, 2, 1), dev)
    dxf = torch.empty(N, C, H, H, device=dev)
    _lib.check(L.lsps_conv2d_dgrad(dy.data_ptr(), w.data_ptr(), dxf.data_ptr(), N, C, H, H, K, 3, 3, 2, 1, wsf, wsfb, st), 'dgradf')
    e_df = float((dxf.double().cpu() - dx_ref).abs().max()) / am
    # weight gradient
    dw_ref = torch.nn.grad.conv2d_weight(x.double().cpu(), (K, C, 3, 3), dy.double().cpu(), stride=2, padding=1)
    dw = torch.empty(K, C, 3, 3, device=dev)
    ws, wsb = _lib.workspace(L.lsps_x3_conv3x3s2_workspace_bytes(N, C, H, H, K), dev)
    _lib.check(L.lsps_x3_conv3x3s2_wgrad(xl.data_ptr(), dyl.data_ptr(), dw.data_ptr(), N, C, H, H, K, ws, wsb, st), 'wgrad')
    e_w = float((dw.double().cpu() - dw_ref).abs().max() / dw_ref.abs().max())
    dwf = torch.empty(K, C, 3, 3, device=dev)
    wsf, wsfb = _lib.workspace(L.lsps_conv2d_workspace_bytes(N, C, H, H, K, 3, 3, 2, 1), dev)
    _lib.check(L.lsps_conv2d_wgrad(x.data_ptr(), dy.data_ptr(), dwf.data_ptr(), None, N, C, H, H, K, 3, 3, 2, 1, wsf, wsfb, st), 'wgradf')
    e_wf = float((dwf.double().cpu() - dw_ref).abs().max() / dw_ref.abs().max())
    print('N=%3d C=%4d %3dx%-3d K=%4d  dgrad x3 %.2e (X3 out %.2e, masked %.2e / f32 out %.2e, db %.2e) f32 kernel %.2e | wgrad x3 %.2e f32 kernel %.2e'
          % (N, C, H, H, K, e_d, e_d3, e_m, e_m32, e_db, e_df, e_w, e_wf), flush=True)

print('== timing (ms per launch; TF = algorithmic 2 N K P Q C 9 / time)')
NG = int(os.environ.get('X3_NGEN', '256'))
ND = int(os.environ.get('X3_NDIS', '768'))
LAYERS = [('gen down1', NG, 64, 128, 128), ('gen down2', NG, 128, 64, 256), ('dis front2', ND, 64, 64, 128),
          ('dis trunk1', ND, 128, 32, 256), ('dis trunk2', ND, 256, 16, 512), ('dis trunk3', ND, 512, 8, 1024),
          ('dis trunk4', ND, 1024, 4, 2048)]
for name, N, C, H, K in LAYERS:
    P = H // 2
    x = torch.randn(N, C, H, H, device=dev)
    w = torch.randn(K, C, 3, 3, device=dev) * 0.05
    b = torch.randn(K, device=dev)
    fl = 2.0 * N * K * P * P * C * 9
    xl = torch.empty(x3_bytes(N, C, H * H), dtype=torch.uint8, device=dev)
    yl = torch.empty(x3_bytes(N, K, P * P), dtype=torch.uint8, device=dev)
    y = torch.empty(N, K, P, P, device=dev)
    assert L.lsps_x3_conv3x3s2_ok(N, C, H, H, K)
    ws, wsb = _lib.workspace(max(L.lsps_x3_conv3x3s2_workspace_bytes(N, C, H, H, K), L.lsps_conv2d_workspace_bytes(N, C, H, H, K, 3, 3, 2, 1),
                                 L.lsps_conv3x3s2_chwn_workspace_bytes(N, C, H, H, K)), dev)
    t_split = timeit(lambda: _lib.check(L.lsps_x3_split_nchw(x.data_ptr(), xl.data_ptr(), N, C, H * H, st), 'split'))
    t_x3 = timeit(lambda: _lib.check(L.lsps_x3_conv3x3s2_fwd(xl.data_ptr(), w.data_ptr(), b.data_ptr(), y.data_ptr(), None, N, C, H, H, K, 0.01, ws, wsb, st), 'x3'))
    t_x3o = timeit(lambda: _lib.check(L.lsps_x3_conv3x3s2_fwd(xl.data_ptr(), w.data_ptr(), b.data_ptr(), None, yl.data_ptr(), N, C, H, H, K, 0.01, ws, wsb, st), 'x3o'))
    t_f = timeit(lambda: _lib.check(L.lsps_conv2d_fwd(x.data_ptr(), w.data_ptr(), b.data_ptr(), y.data_ptr(), N, C, H, H, K, 3, 3, 2, 1, _lib.ACT_LRELU, 0.01, ws, wsb, st), 'f32'))
    kern = L.lsps_last_kernel(st)
    kern = kern.decode() if kern else '?'
    dy = torch.randn(N, K, P, P, device=dev)
    dyl = x3_split(dy)
    dx = torch.empty(N, C, H, H, device=dev)
    dxl = torch.empty(x3_bytes(N, C, H * H), dtype=torch.uint8, device=dev)
    dw = torch.empty(K, C, 3, 3, device=dev)
    db = torch.empty(C, device=dev)
    t_d = timeit(lambda: _lib.check(L.lsps_x3_conv3x3s2_dgrad(dyl.data_ptr(), w.data_ptr(), dx.data_ptr(), None, None, 0.0, None, N, C, H, H, K, ws, wsb, st), 'd'))
    t_dm = timeit(lambda: _lib.check(L.lsps_x3_conv3x3s2_dgrad(dyl.data_ptr(), w.data_ptr(), None, dxl.data_ptr(), xl.data_ptr(), 0.01, db.data_ptr(), N, C, H, H, K, ws, wsb, st), 'dm'))
    t_w = timeit(lambda: _lib.check(L.lsps_x3_conv3x3s2_wgrad(xl.data_ptr(), dyl.data_ptr(), dw.data_ptr(), N, C, H, H, K, ws, wsb, st), 'w'))
    t_df = timeit(lambda: _lib.check(L.lsps_conv2d_dgrad(dy.data_ptr(), w.data_ptr(), dx.data_ptr(), N, C, H, H, K, 3, 3, 2, 1, ws, wsb, st), 'df'))
    t_wf = timeit(lambda: _lib.check(L.lsps_conv2d_wgrad(x.data_ptr(), dy.data_ptr(), dw.data_ptr(), None, N, C, H, H, K, 3, 3, 2, 1, ws, wsb, st), 'wf'))
    extra = ''
    if name.startswith('dis trunk'):
        xc = torch.randn(C, H, H, N, device=dev)
        yc = torch.empty(K, P, P, N, device=dev)
        t_c = timeit(lambda: _lib.check(L.lsps_conv3x3s2_chwn_fwd(xc.data_ptr(), w.data_ptr(), b.data_ptr(), yc.data_ptr(), N, C, H, H, K, _lib.ACT_LRELU, 0.01, ws, wsb, st), 'chwn'))
        dyc = torch.randn(K, P, P, N, device=dev)
        dxc = torch.empty(C, H, H, N, device=dev)
        t_cd = timeit(lambda: _lib.check(L.lsps_conv3x3s2_chwn_dgrad(dyc.data_ptr(), w.data_ptr(), dxc.data_ptr(), N, C, H, H, K, ws, wsb, st), 'chwnd'))
        t_cw = timeit(lambda: _lib.check(L.lsps_conv3x3s2_chwn_wgrad(xc.data_ptr(), dyc.data_ptr(), dw.data_ptr(), N, C, H, H, K, ws, wsb, st), 'chwnw'))
        extra = ' | chwn f32 fwd %.3f dgrad %.3f wgrad %.3f ms' % (t_c, t_cd, t_cw)
    print('%-11s N=%4d C=%4d %3dx%-3d K=%4d %6.1f GFLOP | split %.3f | x3 %.3f ms %4.0f TF | x3 (X3 out) %.3f ms %4.0f TF | f32 %s %.3f ms %4.0f TF%s'
          ' | x3/f32 time %.2f (with split %.2f)\n            dgrad x3 %.3f ms %4.0f TF (masked, X3 out %.3f) f32 %.3f | wgrad x3 %.3f ms %4.0f TF f32 %.3f'
          % (name, N, C, H, H, K, fl / 1e9, t_split, t_x3, fl / t_x3 / 1e9, t_x3o, fl / t_x3o / 1e9, kern, t_f, fl / t_f / 1e9, extra,
             t_x3o / t_f, (t_x3 + t_split) / t_f, t_d, fl / t_d / 1e9, t_dm, t_df, t_w, fl / t_w / 1e9, t_wf), flush=True)
