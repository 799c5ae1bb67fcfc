"""bf16 stride-2 3x3 convs / transposed convs on C8 tensors (csrc/c8s2.h; BASELINE config 5), through the C-ABI.

Every entry against an f64 reference computed from the SAME bf16-rounded operands (as tests/test_c8_gpu.py): the bound
isolates the kernel (f32 accumulation order + one bf16 rounding of the result).  Geometries: every stride-2 layer of the
shipped nets (generator down / up-sampling, discriminator front conv and trunk: 64x64 ... 2x2 output maps) with batch
sizes that leave ragged last tiles (tiles of 4 / 16 / 32 / 64 whole images on the small maps)."""
import os

import pytest
import torch
import torch.nn.functional as F

from test_c8_gpu import BF, C8_TOL, _env, _from_c8, _need_gpu, _rand, _rb, _rel, _to_c8

pytestmark = pytest.mark.gpu

# (N, C, H = W of the conv INPUT, K)
CONV_GEOMS = [(3, 64, 128, 128),        # generator down 1            (lsps_nets.py:186-188)
              (3, 128, 64, 256),        # generator down 2
              (5, 64, 64, 128),         # discriminator front conv 2   (lsps_nets.py:121-123)
              (5, 128, 32, 256),        # trunk layer 1 .. 4           (lsps_nets.py:131-133)
              (6, 256, 16, 512),
              (21, 512, 8, 1024),
              (70, 1024, 4, 2048),
              (1, 64, 16, 128),
              (18, 64, 128, 128)]       # 288 pixel tiles: the persistent workgroups walk 1.125 rounds (ragged last round)
# (N, Ci, H = W of the transposed conv INPUT, Co)
CONVT_GEOMS = [(3, 256, 32, 128), (3, 128, 64, 64), (5, 128, 4, 64), (19, 128, 2, 64),
               (18, 128, 64, 64)]       # 288 pixel tiles of the persistent kernels


def _ws(L, _lib, dev, N, C, H, K):
    return _lib.workspace(L.lsps_c8_conv3x3s2_workspace_bytes(N, C, H, H, K), dev)


@pytest.mark.parametrize("N,C,H,K", CONV_GEOMS)
def test_c8_conv3x3s2_forward_dgrad_wgrad(N, C, H, K):
    _need_gpu()
    _lib, L, dev, st = _env()
    assert L.lsps_c8_conv3x3s2_ok(N, C, H, H, K) == 1
    g = torch.Generator().manual_seed(N * 7 + C + H + K)
    x = _rand(g, N, C, H, H)
    w = _rand(g, K, C, 3, 3, scale=1.0 / (3.0 * C ** 0.5))
    b = _rand(g, K, scale=0.5)
    P = H // 2
    dy = _rand(g, N, K, P, P)
    xd, wd, bd, dyd = _rb(x).double().cpu(), _rb(w).double().cpu(), b.double().cpu(), _rb(dy).double().cpu()
    xc, dyc = _to_c8(x), _to_c8(dy)
    ws, wsb = _ws(L, _lib, dev, N, C, H, K)

    y = torch.empty((N, K // 8, P, P, 8), dtype=BF, device=dev)
    for slope in (0.01, -1.0):
        y.fill_(7.0)
        _lib.check(L.lsps_c8_conv3x3s2_fwd(xc.data_ptr(), w.data_ptr(), b.data_ptr(), y.data_ptr(), N, C, H, H, K, slope, ws, wsb, st), 'fwd')
        ref = F.conv2d(xd, wd, bd, stride=2, padding=1)
        if slope >= 0:
            ref = F.leaky_relu(ref, slope)
        assert _rel(_from_c8(y), ref) <= C8_TOL, ('fwd', slope)
    _lib.check(L.lsps_c8_conv3x3s2_fwd(xc.data_ptr(), w.data_ptr(), None, y.data_ptr(), N, C, H, H, K, -1.0, ws, wsb, st), 'fwd nobias')
    assert _rel(_from_c8(y), F.conv2d(xd, wd, None, stride=2, padding=1)) <= C8_TOL

    dx = torch.full((N, C // 8, H, H, 8), 7.0, dtype=BF, device=dev)
    _lib.check(L.lsps_c8_conv3x3s2_dgrad(dyc.data_ptr(), w.data_ptr(), dx.data_ptr(), N, C, H, H, K, ws, wsb, st), 'dgrad')
    ref = F.conv_transpose2d(dyd, wd, stride=2, padding=1, output_padding=1)
    assert _rel(_from_c8(dx), ref) <= C8_TOL, 'dgrad'

    dw = torch.full((K, C, 3, 3), 7.0, device=dev)
    _lib.check(L.lsps_c8_conv3x3s2_wgrad(xc.data_ptr(), dyc.data_ptr(), dw.data_ptr(), N, C, H, H, K, ws, wsb, st), 'wgrad')
    ref = torch.nn.grad.conv2d_weight(xd, (K, C, 3, 3), dyd, stride=2, padding=1)
    assert _rel(dw, ref) <= 2e-5 * max(1.0, (N * P * P) ** 0.5 / 16), 'wgrad'     # f32 accumulation of exact bf16 products


@pytest.mark.parametrize("N,Ci,H,Co", CONVT_GEOMS)
def test_c8_convT3x3s2_forward_dgrad_wgrad(N, Ci, H, Co):
    _need_gpu()
    _lib, L, dev, st = _env()
    assert L.lsps_c8_convT3x3s2_ok(N, Ci, H, H, Co) == 1
    g = torch.Generator().manual_seed(N * 11 + Ci + H + Co)
    x = _rand(g, N, Ci, H, H)
    w = _rand(g, Ci, Co, 3, 3, scale=1.0 / (1.5 * Ci ** 0.5))
    b = _rand(g, Co, scale=0.5)
    Ho = 2 * H
    dy = _rand(g, N, Co, Ho, Ho)
    xd, wd, bd, dyd = _rb(x).double().cpu(), _rb(w).double().cpu(), b.double().cpu(), _rb(dy).double().cpu()
    xc, dyc = _to_c8(x), _to_c8(dy)
    ws, wsb = _lib.workspace(L.lsps_c8_conv3x3s2_workspace_bytes(N, Co, Ho, Ho, Ci), dev)

    y = torch.full((N, Co // 8, Ho, Ho, 8), 7.0, dtype=BF, device=dev)
    _lib.check(L.lsps_c8_convT3x3s2_fwd(xc.data_ptr(), w.data_ptr(), b.data_ptr(), y.data_ptr(), N, Ci, H, H, Co, 0.01, ws, wsb, st), 'fwd')
    ref = F.leaky_relu(F.conv_transpose2d(xd, wd, bd, stride=2, padding=1, output_padding=1), 0.01)
    assert _rel(_from_c8(y), ref) <= C8_TOL, 'convT fwd'

    dx = torch.full((N, Ci // 8, H, H, 8), 7.0, dtype=BF, device=dev)
    _lib.check(L.lsps_c8_convT3x3s2_dgrad(dyc.data_ptr(), w.data_ptr(), dx.data_ptr(), N, Ci, H, H, Co, ws, wsb, st), 'dgrad')
    assert _rel(_from_c8(dx), F.conv2d(dyd, wd, None, stride=2, padding=1)) <= C8_TOL, 'convT dgrad'

    dw = torch.full((Ci, Co, 3, 3), 7.0, device=dev)
    _lib.check(L.lsps_c8_convT3x3s2_wgrad(xc.data_ptr(), dyc.data_ptr(), dw.data_ptr(), N, Ci, H, H, Co, ws, wsb, st), 'wgrad')
    ref = torch.nn.grad.conv2d_weight(dyd, (Ci, Co, 3, 3), xd, stride=2, padding=1)      # the transposed conv's adjoint roles
    assert _rel(dw, ref) <= 2e-5 * max(1.0, (N * H * H) ** 0.5 / 16), 'convT wgrad'


@pytest.mark.parametrize("N,C,HW", [(3, 64, 4096), (70, 2048, 4), (5, 128, 1024), (1, 8, 5)])
def test_c8_act_bwd_bias(N, C, HW):
    _need_gpu()
    _lib, L, dev, st = _env()
    g = torch.Generator().manual_seed(N + C + HW)
    y = F.leaky_relu(torch.randn(N, C, HW, 1, generator=g), 0.01).cuda()
    y[0, 0, 0, 0] = 0.0                                       # LeakyReLU'(0) = slope, as torch's `out > 0` test
    dy = _rand(g, N, C, HW, 1)
    yc, dyc = _to_c8(y), _to_c8(dy)
    gc = torch.empty_like(dyc)
    db = torch.full((C,), 7.0, device=dev)
    ws, wsb = _lib.workspace(L.lsps_c8_act_bwd_bias_workspace_bytes(N, C), dev)
    _lib.check(L.lsps_c8_act_bwd_bias(dyc.data_ptr(), yc.data_ptr(), gc.data_ptr(), db.data_ptr(), N, C, HW, 0.01, ws, wsb, st), 'abb')
    yd, dyd = _rb(y).double().cpu(), _rb(dy).double().cpu()
    ref = torch.where(yd > 0, dyd, dyd * 0.01)
    got = _from_c8(gc)
    assert _rel(got, ref) <= C8_TOL
    assert _rel(db, got.double().cpu().sum((0, 2, 3))) <= 1e-5 * max(1.0, (N * HW) ** 0.5 / 8)


def test_c8s2_entries_reject_what_they_cannot_do():
    _need_gpu()
    _lib, L, dev, st = _env()
    assert L.lsps_c8_conv3x3s2_ok(4, 64, 48, 48, 128) == 0        # not a power of two
    assert L.lsps_c8_conv3x3s2_ok(4, 24, 32, 32, 128) == 0        # C % 64
    assert L.lsps_c8_conv3x3s2_ok(4, 64, 32, 32, 64) == 0         # K % 128
    assert L.lsps_c8_convT3x3s2_ok(4, 64, 32, 32, 64) == 0        # Ci % 128
    x = torch.zeros(8, dtype=BF, device=dev)
    w = torch.zeros(8, device=dev)
    rc = L.lsps_c8_conv3x3s2_fwd(x.data_ptr(), w.data_ptr(), None, x.data_ptr(), 4, 24, 32, 32, 128, 0.01, None, 0, st)
    assert rc != 0 and b'unsupported geometry' in L.lsps_last_error()


@pytest.mark.parametrize("N,H,stride", [(3, 128, 1), (5, 128, 2), (2, 64, 2)])
def test_c8_stem_forward_and_weight_gradient(N, H, stride):
    """7x7 single-input-channel stems (lsps_nets.py:117,184): C8 output; weight + bias gradient with the LeakyReLU backward
    applied while dy is staged, against f64 from the same bf16-rounded dy / y."""
    _need_gpu()
    _lib, L, dev, st = _env()
    K, R, pad = 64, 7, 3
    assert L.lsps_c8_stem_ok(N, H, H, K, R, R, stride, pad) == 1
    g = torch.Generator().manual_seed(N + H + stride)
    x = _rand(g, N, 1, H, H)
    w = _rand(g, K, 1, R, R, scale=0.1)
    b = _rand(g, K, scale=0.3)
    P = (H + 2 * pad - R) // stride + 1
    y = torch.full((N, K // 8, P, P, 8), 7.0, dtype=BF, device=dev)
    _lib.check(L.lsps_c8_stem_fwd(x.data_ptr(), w.data_ptr(), b.data_ptr(), y.data_ptr(), N, H, H, K, R, R, stride, pad, 0.01, st), 'stem fwd')
    # K = 64 stems run on the bf16 matrix pipe (csrc/c8stem.h): image and weights are rounded to bf16 when the fragments are formed
    xr, wr = (_rb(x), _rb(w)) if os.environ.get('LSPS_C8_STEM_BF16', '1') != '0' else (x, w)
    ref = F.leaky_relu(F.conv2d(xr.double().cpu(), wr.double().cpu(), b.double().cpu(), stride=stride, padding=pad), 0.01)
    assert _rel(_from_c8(y), ref) <= C8_TOL
    dy = _rand(g, N, K, P, P)
    dyc = _to_c8(dy)
    dw = torch.full((K, 1, R, R), 7.0, device=dev)
    db = torch.full((K,), 7.0, device=dev)
    ws, wsb = _lib.workspace(L.lsps_c8_stem_workspace_bytes(K, R, R), dev)
    _lib.check(L.lsps_c8_stem_wgrad(x.data_ptr(), dyc.data_ptr(), y.data_ptr(), dw.data_ptr(), db.data_ptr(), N, H, H, K, R, R, stride, pad,
                                    0.01, ws, wsb, st), 'stem wgrad')
    yd, dyd = _from_c8(y).double().cpu(), _rb(dy).double().cpu()
    gd = torch.where(yd > 0, dyd, dyd * 0.01)
    wref = torch.nn.grad.conv2d_weight(xr.double().cpu(), (K, 1, R, R), gd, stride=stride, padding=pad)
    assert _rel(dw, wref) <= 1e-4
    assert _rel(db, gd.sum((0, 2, 3))) <= 1e-4
    # input gradient (the discriminator stems inside gen_update): bf16 tap GEMM + in-LDS col2im
    if L.lsps_c8_stem_dgrad_ok(N, H, H, K, R, R, stride, pad) == 1:
        dx = torch.full((N, 1, H, H), 7.0, device=dev)
        for rep in range(2):                                  # deterministic: the same bits twice
            _lib.check(L.lsps_c8_stem_dgrad(dyc.data_ptr(), y.data_ptr(), w.data_ptr(), dx.data_ptr(), N, H, H, K, R, R, stride, pad, 0.01,
                                            st), 'stem dgrad')
            if rep == 0:
                first = dx.clone()
        assert torch.equal(first, dx)
        gq = torch.where(yd > 0, dyd, _rb((dyd * 0.01).float()).double())     # the kernel rounds the masked gradient to bf16
        ref = F.conv_transpose2d(gq, _rb(w).double().cpu(), stride=stride, padding=pad, output_padding=H + 2 * pad - R - (P - 1) * stride)
        assert ref.shape == dx.shape
        assert _rel(dx, ref) <= 1e-4


@pytest.mark.parametrize("N,C,H", [(3, 64, 128), (5, 16, 8)])
def test_c8_pw1_head(N, C, H):
    """ConvTranspose2d(C, 1, 1) + Tanh on a C8 input (lsps_nets.py:226-229)."""
    _need_gpu()
    _lib, L, dev, st = _env()
    g = torch.Generator().manual_seed(N + C + H)
    x = _rand(g, N, C, H, H)
    w = _rand(g, C, 1, 1, 1, scale=0.2)
    b = _rand(g, 1, scale=0.3)
    xc = _to_c8(x)
    y = torch.full((N, 1, H, H), 7.0, device=dev)
    _lib.check(L.lsps_c8_pw1_fwd(xc.data_ptr(), w.data_ptr(), b.data_ptr(), y.data_ptr(), N, C, H * H, 2, 0.0, st), 'pw1 fwd')      # 2 = tanh
    xd, wd = _rb(x).double().cpu(), w.double().cpu()
    ref = torch.tanh(F.conv_transpose2d(xd, wd, b.double().cpu()))
    assert _rel(y, ref) <= 1e-5
    dpre = _rand(g, N, 1, H, H)
    dx = torch.full((N, C // 8, H, H, 8), 7.0, dtype=BF, device=dev)
    _lib.check(L.lsps_c8_pw1_dgrad(dpre.data_ptr(), w.data_ptr(), dx.data_ptr(), N, C, H * H, st), 'pw1 dgrad')
    assert _rel(_from_c8(dx), F.conv2d(dpre.double().cpu(), wd)) <= C8_TOL
    dw = torch.full((C, 1, 1, 1), 7.0, device=dev)
    db = torch.full((1,), 7.0, device=dev)
    ws, wsb = _lib.workspace(L.lsps_c8_pw1_workspace_bytes(N, C), dev)
    _lib.check(L.lsps_c8_pw1_wgrad(xc.data_ptr(), dpre.data_ptr(), dw.data_ptr(), db.data_ptr(), N, C, H * H, ws, wsb, st), 'pw1 wgrad')
    assert _rel(dw.view(C), (xd * dpre.double().cpu()).sum((0, 2, 3))) <= 1e-4
    assert _rel(db, dpre.double().cpu().sum().view(1)) <= 1e-4


@pytest.mark.parametrize("N,C,H,K", [(3, 64, 128, 128), (5, 128, 32, 256), (21, 512, 8, 1024), (70, 1024, 4, 2048), (18, 64, 128, 128)])
def test_c8_conv3x3s2_dgrad_with_fused_previous_activation(N, C, H, K):
    """conv dgrad whose epilogue applies the PREVIOUS layer's LeakyReLU backward and sums that layer's bias gradient."""
    _need_gpu()
    _lib, L, dev, st = _env()
    g = torch.Generator().manual_seed(N * 3 + C + H + K)
    w = _rand(g, K, C, 3, 3, scale=1.0 / (3.0 * C ** 0.5))
    P = H // 2
    dy = _rand(g, N, K, P, P)
    yprev = F.leaky_relu(torch.randn(N, C, H, H, generator=g), 0.01).cuda()
    dyc, yc = _to_c8(dy), _to_c8(yprev)
    ws, wsb = _ws(L, _lib, dev, N, C, H, K)
    dx = torch.full((N, C // 8, H, H, 8), 7.0, dtype=BF, device=dev)
    db = torch.full((C,), 7.0, device=dev)
    _lib.check(L.lsps_c8_conv3x3s2_dgrad_act(dyc.data_ptr(), w.data_ptr(), yc.data_ptr(), 0.01, dx.data_ptr(), db.data_ptr(), N, C, H, H, K,
                                             ws, wsb, st), 'dgrad_act')
    ref = F.conv_transpose2d(_rb(dy).double().cpu(), _rb(w).double().cpu(), stride=2, padding=1, output_padding=1)
    ref = torch.where(_rb(yprev).double().cpu() > 0, ref, ref * 0.01)
    got = _from_c8(dx)
    assert _rel(got, ref) <= C8_TOL
    assert _rel(db, got.double().cpu().sum((0, 2, 3))) <= 1e-5 * max(1.0, (N * H * H) ** 0.5 / 8)


@pytest.mark.parametrize("N,Ci,H,Co", [(3, 256, 32, 128), (3, 128, 64, 64), (19, 128, 2, 64), (18, 128, 64, 64)])
def test_c8_convT3x3s2_dgrad_with_fused_previous_activation(N, Ci, H, Co):
    _need_gpu()
    _lib, L, dev, st = _env()
    g = torch.Generator().manual_seed(N * 5 + Ci + H + Co)
    w = _rand(g, Ci, Co, 3, 3, scale=1.0 / (1.5 * Ci ** 0.5))
    Ho = 2 * H
    dy = _rand(g, N, Co, Ho, Ho)
    yprev = F.leaky_relu(torch.randn(N, Ci, H, H, generator=g), 0.01).cuda()
    dyc, yc = _to_c8(dy), _to_c8(yprev)
    ws, wsb = _lib.workspace(L.lsps_c8_conv3x3s2_workspace_bytes(N, Co, Ho, Ho, Ci), dev)
    dx = torch.full((N, Ci // 8, H, H, 8), 7.0, dtype=BF, device=dev)
    db = torch.full((Ci,), 7.0, device=dev)
    _lib.check(L.lsps_c8_convT3x3s2_dgrad_act(dyc.data_ptr(), w.data_ptr(), yc.data_ptr(), 0.01, dx.data_ptr(), db.data_ptr(), N, Ci, H, H,
                                              Co, ws, wsb, st), 'convT dgrad_act')
    ref = F.conv2d(_rb(dy).double().cpu(), _rb(w).double().cpu(), None, stride=2, padding=1)
    ref = torch.where(_rb(yprev).double().cpu() > 0, ref, ref * 0.01)
    got = _from_c8(dx)
    assert _rel(got, ref) <= C8_TOL
    assert _rel(db, got.double().cpu().sum((0, 2, 3))) <= 1e-5 * max(1.0, (N * H * H) ** 0.5 / 8)


@pytest.mark.parametrize('N,C,H', [(70, 64, 32), (5, 64, 20), (3, 16, 128), (9, 40, 11)])
def test_c8_pw1_dgrad_with_fused_previous_activation(N, C, H):
    _need_gpu()
    _lib, L, dev, st = _env()
    g = torch.Generator().manual_seed(9)
    w = _rand(g, C, 1, 1, 1, scale=0.2)
    dpre = _rand(g, N, 1, H, H)
    yprev = F.leaky_relu(torch.randn(N, C, H, H, generator=g), 0.01).cuda()
    yc = _to_c8(yprev)
    dx = torch.full((N, C // 8, H, H, 8), 7.0, dtype=BF, device=dev)
    db = torch.full((C,), 7.0, device=dev)
    ws, wsb = _lib.workspace(L.lsps_c8_pw1_dgrad_act_workspace_bytes(N, C), dev)
    dwh, dbh = torch.full((C, 1, 1, 1), 7.0, device=dev), torch.full((1,), 7.0, device=dev)
    _lib.check(L.lsps_c8_pw1_dgrad_act(dpre.data_ptr(), w.data_ptr(), yc.data_ptr(), 0.01, dx.data_ptr(), db.data_ptr(), dwh.data_ptr(),
                                       dbh.data_ptr(), N, C, H * H, ws, wsb, st), 'pw1 dgrad_act')
    ref = F.conv2d(dpre.double().cpu(), w.double().cpu())
    ref = torch.where(_rb(yprev).double().cpu() > 0, ref, ref * 0.01)
    got = _from_c8(dx)
    assert _rel(got, ref) <= C8_TOL
    assert _rel(db, got.double().cpu().sum((0, 2, 3))) <= 1e-5 * max(1.0, (N * H * H) ** 0.5 / 8)
    # the head's own gradients from the same pass (its input is yprev)
    assert _rel(dwh.view(C), (_rb(yprev).double().cpu() * dpre.double().cpu()).sum((0, 2, 3))) <= 1e-4
    assert _rel(dbh, dpre.double().cpu().sum().view(1)) <= 1e-4


def test_fused_activation_backward_equals_the_separate_pass(monkeypatch):
    """Decoder tail + encoder head of SharedResGen on the C8 kernels (run_layers): gradients with the LeakyReLU backward fused
    into the consumers' dgrad epilogues against the same chain with the separate pass (LSPS_C8_FUSE_ACT=0)."""
    _need_gpu()
    from lsps_amd import ops
    from lsps_amd.trainers import common_net as cn
    torch.manual_seed(3)
    dev = torch.device('cuda')
    dec = [cn.LeakyReLUConvTranspose2d(256, 128, 3, 2, 1, 1), cn.LeakyReLUConvTranspose2d(128, 64, 3, 2, 1, 1),
           cn.ConvTranspose2d(64, 1, 1, 1, 0, act=cn.ACT_TANH)]
    enc = [cn.LeakyReLUConv2d(1, 64, 7, 1, 3), cn.LeakyReLUConv2d(64, 128, 3, 2, 1), cn.LeakyReLUConv2d(128, 256, 3, 2, 1)]
    for m in dec + enc:
        m.to(dev)
    z = torch.randn(3, 256, 32, 32, device=dev)
    ops.set_math_mode('bf16')
    try:
        res = []
        for fuse in ('1', '0'):
            monkeypatch.setattr(ops.options, '_current', ops.options.from_env({'LSPS_C8_FUSE_ACT': fuse}))
            for m in dec + enc:
                for p in m.parameters():
                    p.grad = None
            zz = z.clone().requires_grad_(True)
            out = cn.run_layers(dec, zz)                               # f32 [3, 1, 128, 128]
            lat = ops.from_c8(cn.run_layers(enc, out))
            (lat.square().mean() + out.abs().mean()).backward()
            res.append([zz.grad.clone()] + [p.grad.clone() for m in dec + enc for p in m.parameters()])
    finally:
        ops.set_math_mode('f32')
    for a, b in zip(*res):
        assert float((a - b).abs().max()) <= 2e-2 * float(b.abs().max()) + 1e-12, (a.shape, float((a - b).abs().max()), float(b.abs().max()))
    assert any(not torch.equal(a, b) for a, b in zip(*res))            # the fused path really ran
