#!/usr/bin/env python
"""Numerics + timing of the bf16 C8 residual-conv kernels (csrc/c8conv.h) against f64 references built from the SAME
bf16-rounded operands (so the comparison isolates the kernel: f32 accumulation order + bf16 rounding of the output)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import torch.nn.functional as F  # noqa: E402
from lsps_amd import _lib  # noqa: E402

L = _lib.lib()
dev = torch.device('cuda')
st = _lib.stream()
BF = torch.bfloat16


def to_c8(x):
    N, C, H, W = x.shape
    y = torch.empty((N, C // 8, H, W, 8), dtype=BF, device=x.device)
    _lib.check(L.lsps_c8_from_nchw(x.data_ptr(), y.data_ptr(), N, C, H * W, st), 'from')
    return y


def from_c8(y):
    N, G, H, W, _ = y.shape
    x = torch.empty((N, G * 8, H, W), dtype=torch.float32, device=y.device)
    _lib.check(L.lsps_c8_to_nchw(y.data_ptr(), x.data_ptr(), N, G * 8, H * W, st), 'to')
    return x


def rb(t):
    return t.to(BF).to(torch.float32)


def relerr(a, b):
    return float((a.double() - b.double()).abs().max() / b.double().abs().max())


def check(N, C, K, seed=0):
    g = torch.Generator(device='cpu').manual_seed(seed)
    x = torch.randn(N, C, 32, 32, generator=g).to(dev)
    w = (torch.randn(K, C, 3, 3, generator=g) * (1.0 / (3.0 * C ** 0.5))).to(dev)
    res = torch.randn(N, K, 32, 32, generator=g).to(dev)
    xc = to_c8(x)
    # layout round trip
    assert torch.equal(from_c8(xc), rb(x)), "layout round trip"
    ref_layout = rb(x).view(N, C // 8, 8, 32, 32).permute(0, 1, 3, 4, 2).contiguous()
    assert torch.equal(xc.float(), ref_layout), "layout definition"
    ws, wsb = _lib.workspace(L.lsps_c8_conv3x3_workspace_bytes(C, K), dev)
    xd, wd = rb(x).double().cpu(), rb(w).double().cpu()
    conv = F.conv2d(xd, wd, padding=1)
    out = {}
    # plain forward
    y = torch.empty((N, K // 8, 32, 32, 8), dtype=BF, device=dev)
    _lib.check(L.lsps_c8_conv3x3_fwd(xc.data_ptr(), w.data_ptr(), None, y.data_ptr(), N, C, 32, 32, K, ws, wsb, st), 'fwd')
    out['fwd'] = relerr(from_c8(y).cpu(), conv)
    # conv + IN + LeakyReLU
    rstd = torch.empty(N * K, device=dev)
    a1 = torch.empty_like(y)
    _lib.check(L.lsps_c8_conv3x3_in_fwd(xc.data_ptr(), w.data_ptr(), None, a1.data_ptr(), rstd.data_ptr(), N, C, 32, 32, K, 0.01, 1e-5, ws, wsb, st), 'in1')
    mu = conv.mean((2, 3), keepdim=True)
    var = conv.var((2, 3), unbiased=False, keepdim=True)
    xh = (conv - mu) / (var + 1e-5).sqrt()
    ref1 = F.leaky_relu(xh, 0.01)
    out['in_lrelu'] = relerr(from_c8(a1).cpu(), ref1)
    out['rstd'] = relerr(rstd.cpu().view(N, K), (1.0 / (var + 1e-5).sqrt()).view(N, K))
    # conv + IN + residual
    rc = to_c8(res)
    y2 = torch.empty_like(y)
    _lib.check(L.lsps_c8_conv3x3_in_fwd(xc.data_ptr(), w.data_ptr(), rc.data_ptr(), y2.data_ptr(), rstd.data_ptr(), N, C, 32, 32, K, -1.0, 1e-5, ws, wsb, st), 'in2')
    out['in_res'] = relerr(from_c8(y2).cpu(), xh + rb(res).double().cpu())
    # dgrad (+ addend): dy has K channels, dx has C
    dy = torch.randn(N, K, 32, 32, generator=g).to(dev)
    dyc = to_c8(dy)
    add = torch.randn(N, C, 32, 32, generator=g).to(dev)
    addc = to_c8(add)
    dx = torch.empty((N, C // 8, 32, 32, 8), dtype=BF, device=dev)
    if C % 64 == 0 and K % 16 == 0:
        _lib.check(L.lsps_c8_conv3x3_dgrad_acc(dyc.data_ptr(), w.data_ptr(), addc.data_ptr(), dx.data_ptr(), N, C, 32, 32, K, ws, wsb, st), 'dacc')
        dref = F.conv_transpose2d(rb(dy).double().cpu(), wd, padding=1)
        out['dgrad_acc'] = relerr(from_c8(dx).cpu(), dref + rb(add).double().cpu())
        # dgrad through IN + LeakyReLU backward, from the saved output `o` (C channels) and its rstd
        o = F.leaky_relu(torch.randn(N, C, 32, 32, generator=g), 0.01).to(dev)
        oc = to_c8(o)
        rs = (torch.rand(N * C, generator=g) + 0.5).to(dev)
        _lib.check(L.lsps_c8_conv3x3_dgrad_inbwd(dyc.data_ptr(), w.data_ptr(), oc.data_ptr(), rs.data_ptr(), dx.data_ptr(), N, C, 32, 32, K, 0.01, ws, wsb, st), 'dinb')
        od = rb(o).double().cpu()
        pos = od > 0
        gg = torch.where(pos, dref, dref * 0.01)
        xhh = torch.where(pos, od, od / 0.01)
        m1 = gg.mean((2, 3), keepdim=True)
        m2 = (gg * xhh).mean((2, 3), keepdim=True)
        refb = rs.double().cpu().view(N, C, 1, 1) * (gg - m1 - xhh * m2)
        out['dgrad_inbwd'] = relerr(from_c8(dx).cpu(), refb)
    # weight gradient (K % 128 == 0, C % 64 == 0): dW[k][c] = sum dy * shifted x
    if K % 128 == 0 and C % 64 == 0:
        wsw, wswb = _lib.workspace(L.lsps_c8_conv3x3_wgrad_workspace_bytes(N, C, K), dev)
        dw = torch.empty(K, C, 3, 3, device=dev)
        _lib.check(L.lsps_c8_conv3x3_wgrad(xc.data_ptr(), dyc.data_ptr(), dw.data_ptr(), N, C, 32, 32, K, wsw, wswb, st), 'wgrad')
        xg = xd.clone().requires_grad_(False)
        wref = torch.nn.grad.conv2d_weight(xd, (K, C, 3, 3), rb(dy).double().cpu(), padding=1)
        out['wgrad'] = relerr(dw.cpu(), wref)
    # InstanceNorm (+ residual) backward from the output
    gq, oq, rq = (torch.randn(N, K, 32, 32, generator=g).to(dev) for _ in range(3))
    rs = (torch.rand(N * K, generator=g) + 0.5).to(dev)
    gc, oc2, rcc = to_c8(gq), to_c8(oq), to_c8(rq)
    dyo = torch.empty_like(gc)
    _lib.check(L.lsps_c8_inorm_bwd(gc.data_ptr(), oc2.data_ptr(), rcc.data_ptr(), rs.data_ptr(), dyo.data_ptr(), N, K, 1024, -1.0, st), 'inbwd')
    gd, xhd = rb(gq).double().cpu(), (rb(oq) - rb(rq)).double().cpu()
    refn = rs.double().cpu().view(N, K, 1, 1) * (gd - gd.mean((2, 3), keepdim=True) - xhd * (gd * xhd).mean((2, 3), keepdim=True))
    out['inorm_bwd_res'] = relerr(from_c8(dyo).cpu(), refn)
    _lib.check(L.lsps_c8_inorm_bwd(gc.data_ptr(), oc2.data_ptr(), None, rs.data_ptr(), dyo.data_ptr(), N, K, 1024, 0.01, st), 'inbwd')
    od = rb(oq).double().cpu()
    pos = od > 0
    g2 = torch.where(pos, gd, gd * 0.01)
    xh2 = torch.where(pos, od, od / 0.01)
    refn = rs.double().cpu().view(N, K, 1, 1) * (g2 - g2.mean((2, 3), keepdim=True) - xh2 * (g2 * xh2).mean((2, 3), keepdim=True))
    out['inorm_bwd_act'] = relerr(from_c8(dyo).cpu(), refn)
    return out


def timing(N, C=256, K=256, reps=5):
    x = torch.randn(N, C, 32, 32, device=dev)
    w = torch.randn(K, C, 3, 3, device=dev) * 0.02
    xc = to_c8(x)
    y = torch.empty((N, K // 8, 32, 32, 8), dtype=BF, device=dev)
    rstd = torch.empty(N * K, device=dev)
    ws, wsb = _lib.workspace(L.lsps_c8_conv3x3_workspace_bytes(C, K), dev)
    fl = 2.0 * N * K * 1024 * C * 9

    def run(fn):
        fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / reps
    t0 = run(lambda: _lib.check(L.lsps_c8_conv3x3_fwd(xc.data_ptr(), w.data_ptr(), None, y.data_ptr(), N, C, 32, 32, K, ws, wsb, st), 'f'))
    t1 = run(lambda: _lib.check(L.lsps_c8_conv3x3_in_fwd(xc.data_ptr(), w.data_ptr(), None, y.data_ptr(), rstd.data_ptr(), N, C, 32, 32, K, 0.01, 1e-5, ws, wsb, st), 'f'))
    t3 = run(lambda: _lib.check(L.lsps_c8_conv3x3_dgrad_inbwd(xc.data_ptr(), w.data_ptr(), y.data_ptr(), rstd.data_ptr(), y.data_ptr(), N, C, 32, 32, K, 0.01, ws, wsb, st), 'd')) if False else 0.0
    tc = run(lambda: to_c8(x))
    wsw, wswb = _lib.workspace(L.lsps_c8_conv3x3_wgrad_workspace_bytes(N, C, K), dev)
    dw = torch.empty(K, C, 3, 3, device=dev)
    tw = run(lambda: _lib.check(L.lsps_c8_conv3x3_wgrad(xc.data_ptr(), y.data_ptr(), dw.data_ptr(), N, C, 32, 32, K, wsw, wswb, st), 'w'))
    td = run(lambda: _lib.check(L.lsps_c8_conv3x3_dgrad_inbwd(xc.data_ptr(), w.data_ptr(), y.data_ptr(), rstd.data_ptr(), xc.data_ptr(), N, C, 32, 32, K, 0.01, ws, wsb, st), 'd'))
    tn = run(lambda: _lib.check(L.lsps_c8_inorm_bwd(xc.data_ptr(), y.data_ptr(), xc.data_ptr(), rstd.data_ptr(), xc.data_ptr(), N, K, 1024, -1.0, st), 'n'))
    print('        wgrad %.3f ms %.0f TF | dgrad+INbwd %.3f ms %.0f TF | inorm_bwd %.3f ms (%.2f TB/s)' %
          (tw, fl / tw / 1e9, td, fl / td / 1e9, tn, N * K * 1024 * 8 / tn / 1e9))
    print('N=%4d  plain %.3f ms %.0f TF | conv+IN %.3f ms %.0f TF | to_c8 %.3f ms (%.2f TB/s)' %
          (N, t0, fl / t0 / 1e9, t1, fl / t1 / 1e9, tc, N * C * 1024 * 6 / tc / 1e9))


if __name__ == '__main__':
    if 'time' not in sys.argv[1:]:
        for (N, C, K) in ((2, 64, 64), (3, 256, 256), (9, 32, 128), (8, 128, 64), (19, 64, 128), (40, 128, 128)):
            print((N, C, K), {k: '%.2e' % v for k, v in check(N, C, K).items()})
    if 'check' not in sys.argv[1:]:
        for N in (64, 256, 512, 768):
            timing(N)
