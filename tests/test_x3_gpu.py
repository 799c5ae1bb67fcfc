"""The three-limb ("X3") 3x3 / stride-2 family of csrc/x3s2.h, kernel by kernel through the C-ABI against f64 CPU references
(reference layers: src/trainers/common_net.py:246-268 LeakyReLUConv2d / LeakyReLUConvTranspose2d as used in lsps_nets.py:117-124,
186-192, 222-225).  The arithmetic is f32-class: the bound is the f32 kernels' (2e-5 of the tensor's abs-max, test_kernels_gpu.py),
measured 1e-6 - 3e-6; the split into limbs is exact."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
TOL = 2e-5
BF = torch.bfloat16


def _need_gpu():
    if not torch.cuda.is_available():
        pytest.skip("no HIP device")


def _rel(a, b):
    return float((a.double().cpu() - b.double().cpu()).abs().max() / b.double().abs().max())


def _env():
    from lsps_amd import _lib, ops
    return _lib, _lib.lib(), ops, torch.device('cuda'), _lib.stream()


# (N, C, H, K): every stride-2 layer geometry of the two nets (generator down 1 / 2, discriminator front 2, trunk 1 - 4) at ragged
# batch sizes (partial image tiles, partial last chunks)
GEOMS = [(2, 64, 128, 128), (3, 128, 64, 256), (3, 64, 64, 128), (5, 128, 32, 256), (7, 256, 16, 512), (19, 512, 8, 1024),
         (70, 1024, 4, 2048)]


def test_split_join_is_exact():
    _need_gpu()
    _lib, L, ops, dev, st = _env()
    torch.manual_seed(0)
    # exact for +0 and for |x| >= 2^-100: below that the lowest limb (2^-16 .. 2^-24 of x) underflows the f32 / bf16 exponent
    # range and bits are lost (1.5e-37 comes back as 1.4997e-37), and -0 comes back as +0: nothing a convolution can tell
    x = torch.randn(3, 64, 16, 16, device=dev) * torch.logspace(-15, 20, 16, device=dev)
    x[0, 0, 0, :4] = torch.tensor([0.0, 1.0, 1.5e-30, 3.0e38], device=dev)
    xl = ops.x3_split(x)
    assert ops.is_x3(xl) and tuple(xl.shape) == (3, 3, 8, 16, 16, 8)
    assert torch.equal(ops.x3_join(xl), x)
    # the hi limb alone is the bf16 rounding of x in the C8 layout
    hi = xl[:, 0].float().permute(0, 1, 4, 2, 3).reshape(3, 64, 16, 16)
    assert torch.equal(hi, x.to(BF).float())


@pytest.mark.parametrize("N,C,H,K", GEOMS)
def test_conv_forward_dgrad_wgrad(N, C, H, K):
    _need_gpu()
    _lib, L, ops, dev, st = _env()
    assert L.lsps_x3_conv3x3s2_ok(N, C, H, H, K) == 1
    torch.manual_seed(N + C)
    P = H // 2
    x = torch.randn(N, C, H, H, device=dev)
    w = torch.randn(K, C, 3, 3, device=dev) * 0.05
    b = torch.randn(K, device=dev)
    dy = torch.randn(N, K, P, P, device=dev)
    yprev = torch.randn(N, C, H, H, device=dev)
    xl, dyl, ypl = ops.x3_split(x), ops.x3_split(dy), ops.x3_split(yprev)
    ws, wsb = _lib.workspace(L.lsps_x3_conv3x3s2_workspace_bytes(N, C, H, H, K), dev)
    xd, wd, dyd = x.double().cpu(), w.double().cpu(), dy.double().cpu()
    # forward, both output forms
    ref = F.leaky_relu(F.conv2d(xd, wd, b.double().cpu(), stride=2, padding=1), 0.01)
    y = torch.empty(N, K, P, P, device=dev)
    _lib.check(L.lsps_x3_conv3x3s2_fwd(_lib.ptr(xl, BF), _lib.ptr(w), _lib.ptr(b), _lib.ptr(y), None, N, C, H, H, K, 0.01, ws, wsb, st), 'f')
    assert _rel(y, ref) < TOL
    yl = torch.empty(N, 3, K // 8, P, P, 8, dtype=BF, device=dev)
    _lib.check(L.lsps_x3_conv3x3s2_fwd(_lib.ptr(xl, BF), _lib.ptr(w), _lib.ptr(b), None, _lib.ptr(yl, BF), N, C, H, H, K, 0.01, ws, wsb, st), 'f3')
    assert torch.equal(ops.x3_join(yl), y)                       # the same accumulators, split exactly
    # no bias, no activation
    _lib.check(L.lsps_x3_conv3x3s2_fwd(_lib.ptr(xl, BF), _lib.ptr(w), None, _lib.ptr(y), None, N, C, H, H, K, -1.0, ws, wsb, st), 'f0')
    assert _rel(y, F.conv2d(xd, wd, None, stride=2, padding=1)) < TOL
    # input gradient: plain (f32 / X3 out) and with the previous layer's LeakyReLU backward + bias gradient fused
    dx_ref = torch.nn.grad.conv2d_input((N, C, H, H), wd, dyd, stride=2, padding=1)
    dx = torch.empty(N, C, H, H, device=dev)
    _lib.check(L.lsps_x3_conv3x3s2_dgrad(_lib.ptr(dyl, BF), _lib.ptr(w), _lib.ptr(dx), None, None, 0.0, None, N, C, H, H, K, ws, wsb, st), 'd')
    assert _rel(dx, dx_ref) < TOL
    dxl = torch.empty(N, 3, C // 8, H, H, 8, dtype=BF, device=dev)
    _lib.check(L.lsps_x3_conv3x3s2_dgrad(_lib.ptr(dyl, BF), _lib.ptr(w), None, _lib.ptr(dxl, BF), None, 0.0, None, N, C, H, H, K, ws, wsb, st), 'd3')
    assert torch.equal(ops.x3_join(dxl), dx)
    g_ref = torch.where(yprev.double().cpu() > 0, dx_ref, dx_ref * 0.01)
    db = torch.empty(C, device=dev)
    _lib.check(L.lsps_x3_conv3x3s2_dgrad(_lib.ptr(dyl, BF), _lib.ptr(w), None, _lib.ptr(dxl, BF), _lib.ptr(ypl, BF), 0.01, _lib.ptr(db), N, C, H, H, K,
                                         ws, wsb, st), 'dm')
    assert _rel(ops.x3_join(dxl), g_ref) < TOL
    assert _rel(db, g_ref.sum((0, 2, 3))) < 1e-4
    _lib.check(L.lsps_x3_conv3x3s2_dgrad(_lib.ptr(dyl, BF), _lib.ptr(w), _lib.ptr(dx), None, _lib.ptr(ypl, BF), 0.01, _lib.ptr(db), N, C, H, H, K,
                                         ws, wsb, st), 'dm32')
    assert _rel(dx, g_ref) < TOL
    # weight gradient
    dw_ref = torch.nn.grad.conv2d_weight(xd, (K, C, 3, 3), dyd, stride=2, padding=1)
    dw = torch.empty(K, C, 3, 3, device=dev)
    _lib.check(L.lsps_x3_conv3x3s2_wgrad(_lib.ptr(xl, BF), _lib.ptr(dyl, BF), _lib.ptr(dw), N, C, H, H, K, ws, wsb, st), 'w')
    assert _rel(dw, dw_ref) < TOL


@pytest.mark.parametrize("N,C,H,K", [(3, 128, 64, 256), (5, 64, 128, 128), (19, 512, 8, 1024)])
def test_forward_ring_variant_is_bitwise_the_double_buffered_kernel(N, C, H, K):
    """Round 6 experiment kept behind `options.x3_ring` (image pieces requested two stages ahead into a ring of
    three buffers, counted stage-end waits; measured 7 - 17 % SLOWER, profiles/r6f_x3_ring_ab.txt, so it is off): the same MFMAs
    in the same order, so every output bit must equal the default kernel's — f32 and X3 output, persistent multi-tile walk."""
    _need_gpu()
    _lib, L, ops, dev, st = _env()
    torch.manual_seed(N)
    P = H // 2
    x = torch.randn(N, C, H, H, device=dev)
    w = torch.randn(K, C, 3, 3, device=dev) * 0.05
    b = torch.randn(K, device=dev)
    xl = ops.x3_split(x)
    ws, wsb = _lib.workspace(L.lsps_x3_conv3x3s2_workspace_bytes(N, C, H, H, K), dev)
    outs = []
    for ring in (False, True):
        with ops.options.override(x3_ring=ring):
            assert _lib.native_options()['x3_ring'] == int(ring)
            y = torch.empty(N, K, P, P, device=dev)
            yl = torch.empty(N, 3, K // 8, P, P, 8, dtype=BF, device=dev)
            ops.kernel_log_begin()
            _lib.check(L.lsps_x3_conv3x3s2_fwd(_lib.ptr(xl, BF), _lib.ptr(w), _lib.ptr(b), _lib.ptr(y), None, N, C, H, H, K, 0.01, ws, wsb, st), 'f')
            _lib.check(L.lsps_x3_conv3x3s2_fwd(_lib.ptr(xl, BF), _lib.ptr(w), _lib.ptr(b), None, _lib.ptr(yl, BF), N, C, H, H, K, 0.01, ws, wsb, st), 'f3')
            ops.kernel_log_end()
            outs.append((y, yl))
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])
    ref = F.leaky_relu(F.conv2d(x.double().cpu(), w.double().cpu(), b.double().cpu(), stride=2, padding=1), 0.01)
    assert _rel(outs[1][0], ref) < TOL


@pytest.mark.parametrize("N,Ci,H,Co", [(3, 256, 32, 128), (2, 128, 64, 64), (5, 256, 8, 128)])
def test_transposed_conv_forward_dgrad_wgrad(N, Ci, H, Co):
    _need_gpu()
    _lib, L, ops, dev, st = _env()
    assert L.lsps_x3_convT3x3s2_ok(N, Ci, H, H, Co) == 1
    torch.manual_seed(N + Ci)
    x = torch.randn(N, Ci, H, H, device=dev)
    w = torch.randn(Ci, Co, 3, 3, device=dev) * 0.05
    b = torch.randn(Co, device=dev)
    dy = torch.randn(N, Co, 2 * H, 2 * H, device=dev)
    xl, dyl = ops.x3_split(x), ops.x3_split(dy)
    ws, wsb = _lib.workspace(L.lsps_x3_conv3x3s2_workspace_bytes(N, Co, 2 * H, 2 * H, Ci), dev)
    xd = x.double().cpu().requires_grad_(True)
    wd = w.double().cpu().requires_grad_(True)
    ref = F.conv_transpose2d(xd, wd, b.double().cpu(), stride=2, padding=1, output_padding=1)
    ref.backward(dy.double().cpu())
    y = torch.empty(N, Co, 2 * H, 2 * H, device=dev)
    _lib.check(L.lsps_x3_convT3x3s2_fwd(_lib.ptr(xl, BF), _lib.ptr(w), _lib.ptr(b), _lib.ptr(y), None, N, Ci, H, H, Co, 0.01, ws, wsb, st), 'f')
    assert _rel(y, F.leaky_relu(ref.detach(), 0.01)) < TOL
    yl = torch.empty(N, 3, Co // 8, 2 * H, 2 * H, 8, dtype=BF, device=dev)
    _lib.check(L.lsps_x3_convT3x3s2_fwd(_lib.ptr(xl, BF), _lib.ptr(w), _lib.ptr(b), None, _lib.ptr(yl, BF), N, Ci, H, H, Co, 0.01, ws, wsb, st), 'f3')
    assert torch.equal(ops.x3_join(yl), y)
    dx = torch.empty(N, Ci, H, H, device=dev)
    _lib.check(L.lsps_x3_convT3x3s2_dgrad(_lib.ptr(dyl, BF), _lib.ptr(w), _lib.ptr(dx), None, None, 0.0, None, N, Ci, H, H, Co, ws, wsb, st), 'd')
    assert _rel(dx, xd.grad) < TOL
    dw = torch.empty_like(w)
    _lib.check(L.lsps_x3_convT3x3s2_wgrad(_lib.ptr(xl, BF), _lib.ptr(dyl, BF), _lib.ptr(dw), N, Ci, H, H, Co, ws, wsb, st), 'w')
    assert _rel(dw, wd.grad) < TOL


def test_transposed_conv_dgrad_with_fused_activation_backward():
    """ConvTranspose2d dgrad (forward-direction kernel) with the LeakyReLU backward + bias gradient of the X3 layer in front."""
    _need_gpu()
    _lib, L, ops, dev, st = _env()
    torch.manual_seed(11)
    N, Ci, H, Co = 3, 256, 16, 128
    w = torch.randn(Ci, Co, 3, 3, device=dev) * 0.05
    dy = torch.randn(N, Co, 2 * H, 2 * H, device=dev)
    yprev = torch.randn(N, Ci, H, H, device=dev)
    dyl, ypl = ops.x3_split(dy), ops.x3_split(yprev)
    ws, wsb = _lib.workspace(L.lsps_x3_conv3x3s2_workspace_bytes(N, Co, 2 * H, 2 * H, Ci), dev)
    ref = F.conv2d(dy.double().cpu(), w.double().cpu(), None, stride=2, padding=1)          # dgrad of the transposed conv
    g_ref = torch.where(yprev.double().cpu() > 0, ref, ref * 0.01)
    dxl = torch.empty(N, 3, Ci // 8, H, H, 8, dtype=BF, device=dev)
    db = torch.empty(Ci, device=dev)
    _lib.check(L.lsps_x3_convT3x3s2_dgrad(_lib.ptr(dyl, BF), _lib.ptr(w), None, _lib.ptr(dxl, BF), _lib.ptr(ypl, BF), 0.01, _lib.ptr(db), N, Ci, H, H,
                                          Co, ws, wsb, st), 'dm')
    assert _rel(ops.x3_join(dxl), g_ref) < TOL
    assert _rel(db, g_ref.sum((0, 2, 3))) < 1e-4
    dx = torch.empty(N, Ci, H, H, device=dev)
    _lib.check(L.lsps_x3_convT3x3s2_dgrad(_lib.ptr(dyl, BF), _lib.ptr(w), _lib.ptr(dx), None, _lib.ptr(ypl, BF), 0.01, _lib.ptr(db), N, Ci, H, H, Co,
                                          ws, wsb, st), 'dm32')
    assert _rel(dx, g_ref) < TOL


@pytest.mark.parametrize("stride", [1, 2])
def test_stem_writes_limbs(stride):
    """The 7x7 one-input-channel stems with the limb-splitting epilogue: exactly the f32 kernel's output, split."""
    _need_gpu()
    _lib, L, ops, dev, st = _env()
    torch.manual_seed(12)
    N, H, K = 3, 128, 64
    x = torch.randn(N, 1, H, H, device=dev)
    w = torch.randn(K, 1, 7, 7, device=dev) * 0.1
    b = torch.randn(K, device=dev)
    assert L.lsps_x3_stem_ok(N, H, H, K, 7, 7, stride, 3) == 1
    P = (H + 6 - 7) // stride + 1
    yl = torch.empty(N, 3, K // 8, P, P, 8, dtype=BF, device=dev)
    _lib.check(L.lsps_x3_stem_fwd(_lib.ptr(x), _lib.ptr(w), _lib.ptr(b), _lib.ptr(yl, BF), N, H, H, K, 7, 7, stride, 3, 0.01, st), 'stem')
    y = ops.conv2d(x, w, b, stride, 3, ops.ACT_LRELU, 0.01)
    assert torch.equal(ops.x3_join(yl), y)
    ref = F.leaky_relu(F.conv2d(x.double().cpu(), w.double().cpu(), b.double().cpu(), stride=stride, padding=3), 0.01)
    assert _rel(y, ref) < TOL


@pytest.mark.parametrize("H", [64, 20, 128])
def test_output_head_dgrad_emits_limbs(H):
    """lsps_pw1_dgrad_act_x3: the 1x1 head's input gradient with the LeakyReLU backward of the layer in front, written as limbs —
    bit for bit the f32 entry's result, split.  (H = 20: 50 pixel quads per block, the kernel's ragged pass; 64 / 128: full
    1024-pixel passes only.)"""
    _need_gpu()
    _lib, L, ops, dev, st = _env()
    torch.manual_seed(13)
    N, C = 3, 64
    dpre, w, y = torch.randn(N, 1, H, H, device=dev), torch.randn(C, device=dev), torch.randn(N, C, H, H, device=dev)
    ws, wsb = _lib.workspace(L.lsps_pw1_dgrad_act_workspace_bytes(N, C), dev)
    dx, db0, dw0 = torch.empty(N, C, H, H, device=dev), torch.empty(C, device=dev), torch.empty(C, device=dev)
    _lib.check(L.lsps_pw1_dgrad_act(_lib.ptr(dpre), _lib.ptr(w), _lib.ptr(y), 0.01, _lib.ptr(dx), _lib.ptr(db0), _lib.ptr(dw0), None, N, C, H * H,
                                    ws, wsb, st), 'pw1')
    gl, db1, dw1 = torch.empty(N, 3, C // 8, H, H, 8, dtype=BF, device=dev), torch.empty(C, device=dev), torch.empty(C, device=dev)
    _lib.check(L.lsps_pw1_dgrad_act_x3(_lib.ptr(dpre), _lib.ptr(w), _lib.ptr(y), 0.01, _lib.ptr(gl, BF), _lib.ptr(db1), _lib.ptr(dw1), None, N, C,
                                       H * H, ws, wsb, st), 'pw1x3')
    assert torch.equal(ops.x3_join(gl), dx)
    # (the two instantiations contract their partial sums differently: round-off, not bitwise)
    assert _rel(db1, db0) < 1e-6 and _rel(dw1, dw0) < 1e-6
    ref = torch.where(y > 0, w.view(1, C, 1, 1) * dpre, w.view(1, C, 1, 1) * dpre * 0.01)
    assert _rel(dx, ref) < 1e-6


def test_activation_backward_emits_limbs():
    _need_gpu()
    _lib, L, ops, dev, st = _env()
    torch.manual_seed(4)
    N, C, H = 5, 128, 16
    dy, y = torch.randn(N, C, H, H, device=dev), torch.randn(N, C, H, H, device=dev)
    y[0, 0, 0, :2] = torch.tensor([0.0, -0.0], device=dev)          # LeakyReLU'(0) = slope, like torch's `out > 0` test
    g = torch.empty(N, 3, C // 8, H, H, 8, dtype=BF, device=dev)
    db = torch.empty(C, device=dev)
    ws, wsb = _lib.workspace(L.lsps_x3_act_bwd_bias_workspace_bytes(N, C), dev)
    _lib.check(L.lsps_x3_act_bwd_bias(_lib.ptr(dy), _lib.ptr(y), _lib.ptr(g, BF), _lib.ptr(db), N, C, H * H, 0.01, ws, wsb, st), 'a')
    ref = torch.where(y > 0, dy, dy * 0.01)
    assert torch.equal(ops.x3_join(g), ref)
    assert _rel(db, ref.double().sum((0, 2, 3))) < 1e-5
    _lib.check(L.lsps_x3_act_bwd_bias(_lib.ptr(dy), None, _lib.ptr(g, BF), None, N, C, H * H, -1.0, ws, wsb, st), 'a0')
    assert torch.equal(ops.x3_join(g), dy)


@pytest.mark.parametrize("what", ["dis_trunk", "encoder_front", "decoder_tail"])
def test_layer_chains_match_the_f32_kernels(what, monkeypatch):
    """run_layers with the three-limb family forced on against the exact-f32 kernels on the same layers: outputs and every
    gradient (inputs, weights, biases) agree to f32 round-off — chaining, fused LeakyReLU backward, f32 hand-over included.
    The forward tensors differ by ~1e-6 relative, so a handful of LeakyReLU masks flip (pre-activations within that distance
    of 0): outputs are compared element by element, gradients by their L2 distance and the MEDIAN element error (one flipped
    mask moves everything in its receptive field — a flip at 128 x 128 reaches ~2000 of the 786 k input-gradient elements of the
    decoder tail — so upper quantiles measure the number of flips, not the arithmetic)."""
    _need_gpu()
    from lsps_amd import ops
    from lsps_amd.trainers import common_net as cn
    torch.manual_seed(9)
    dev = torch.device('cuda')
    if what == 'dis_trunk':
        layers = [cn.LeakyReLUConv2d(128, 256, 3, 2, 1), cn.LeakyReLUConv2d(256, 512, 3, 2, 1), cn.LeakyReLUConv2d(512, 1024, 3, 2, 1),
                  cn.LeakyReLUConv2d(1024, 2048, 3, 2, 1)]
        x = torch.randn(6, 128, 32, 32, device=dev)
    elif what == 'encoder_front':
        layers = [cn.LeakyReLUConv2d(1, 64, 7, 1, 3), cn.LeakyReLUConv2d(64, 128, 3, 2, 1), cn.LeakyReLUConv2d(128, 256, 3, 2, 1)]
        x = torch.randn(3, 1, 128, 128, device=dev)
    else:
        layers = [cn.LeakyReLUConvTranspose2d(256, 128, 3, 2, 1, 1), cn.LeakyReLUConvTranspose2d(128, 64, 3, 2, 1, 1),
                  cn.ConvTranspose2d(64, 1, 1, 1, 0, act=cn.ACT_TANH)]
        x = torch.randn(3, 256, 32, 32, device=dev)
    for m in layers:
        m.to(dev)
    res = []
    probe = None
    # exact-f32 kernels, then the three-limb family; then the family with LSPS_C8_FUSE_ACT=0 (the X3 stem used to pair
    # with its consumer on `fuse_act` alone and its backward raised when the consumer took the unfused branch)
    for gmac, fuse in (('1e9', '1'), ('0', '1'), ('0', '0')):
        monkeypatch.setattr(ops.options, '_current', ops.options.from_env({'LSPS_X3_MIN_GMAC': gmac, 'LSPS_C8_FUSE_ACT': fuse}))
        for m in layers:
            for p in m.parameters():
                p.grad = None
        xx = x.clone().requires_grad_(True)
        ops.kernel_log_begin()
        out = ops.from_c8(cn.run_layers(layers, xx))
        if probe is None:
            probe = torch.randn_like(out)
        (out.square().mean() + (out * probe).mean()).backward()
        names = ops.kernel_log_end()
        assert any(k.startswith('x3s2_') for k in names) == (gmac == '0'), names
        res.append([out.detach().clone(), xx.grad.clone()] + [p.grad.clone() for m in layers for p in m.parameters()])
    for k in (1, 2):
        assert float((res[0][0] - res[k][0]).abs().max()) <= 2e-5 * float(res[0][0].abs().max())
        for i, (a, b) in enumerate(zip(res[k][1:], res[0][1:])):
            d, am = (a - b).abs().double().flatten(), float(b.abs().max())
            q = float(torch.quantile(d[:: max(1, d.numel() // 1000000)], 0.5))
            assert q <= 2e-5 * am + 1e-12, (a.shape, q, am)
            # ONE flipped mask at 128 x 128 changes ~0.3 % of the input gradient's elements by ~3 % each = 1.6e-3 of its L2 norm
            # (measured: 1.4e-3 with one flip); the parameter gradients sum over every pixel of only 3 samples here and see it at ~1e-3
            # (measured 1.15e-3 on the first transposed conv's weight).  A wrong tap, mask or scale would show at 0.1 - 1.
            l2 = 5e-3 if i == 0 else 3e-3
            assert float(d.norm()) <= l2 * float(b.double().norm()) + 1e-12, (a.shape, float(d.norm()), float(b.double().norm()))
