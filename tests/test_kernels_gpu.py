"""Per-kernel parity of liblsps_hip.so (through the C-ABI, via lsps_amd.ops) against plain torch
fp32 CPU ops — the same torch ops the oracle is made of.  Tolerance: 1e-4 relative to the
reference tensor's abs-max for forward / dgrad, 2e-4 for weight gradients (long fp32 reductions).
"""
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def _need_gpu():
    if not torch.cuda.is_available():
        pytest.skip("no HIP device")


def _rel(a, b):
    a, b = a.detach().cpu().double(), b.detach().cpu().double()
    return float((a - b).abs().max() / max(b.abs().max().item(), 1e-30))


def _rand(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return (torch.rand(*shape, generator=g) * 2 - 1) * scale


# (N, C, H, W, K, R, stride, pad) — every conv geometry of SharedResGen / SharedDis (ch=64 and tiny),
# plus ragged sizes that do not fill tiles.
CONV_CASES = [
    (2, 1, 128, 128, 64, 7, 1, 3),      # gen stem
    (2, 64, 128, 128, 128, 3, 2, 1),    # gen down 1
    (2, 128, 64, 64, 256, 3, 2, 1),     # gen down 2
    (3, 256, 32, 32, 256, 3, 1, 1),     # residual conv (dominant)
    (2, 1, 128, 128, 64, 7, 2, 3),      # dis stem
    (2, 64, 64, 64, 128, 3, 2, 1),      # dis front
    (4, 128, 32, 32, 256, 3, 2, 1),     # dis shared 0
    (4, 256, 16, 16, 512, 3, 2, 1),
    (4, 512, 8, 8, 1024, 3, 2, 1),
    (4, 1024, 4, 4, 2048, 3, 2, 1),
    (6, 2048, 2, 2, 1, 1, 1, 0),        # D head
    (6, 2048, 2, 2, 20, 2, 1, 0),       # Post head
    (1, 2048, 2, 2, 20, 2, 1, 0),       # Post head, n = 1
    (2, 8, 128, 128, 16, 3, 2, 1),      # tiny-width layers
    (5, 3, 17, 13, 37, 3, 1, 1),        # ragged everything
    (3, 5, 19, 23, 70, 5, 2, 2),        # 5x5 stride 2, ragged
    (2, 33, 9, 9, 129, 3, 3, 0),        # stride 3
    (2, 8, 32, 32, 130, 3, 1, 1),       # specialised 3x3/width-32 kernel: one channel chunk, ragged M
    (5, 16, 8, 32, 128, 3, 1, 1),       # specialised kernel: H=8 (two row tiles), N not a power of two
    (1, 24, 4, 32, 200, 3, 1, 1),       # specialised kernel: a single row tile (both halos out of range)
    (3, 1, 128, 128, 24, 7, 2, 3),      # direct 1-channel-input dgrad (discriminator stem), K not a multiple of 16
    (2, 1, 64, 128, 16, 7, 2, 3),       # same, non-square image
    (3, 64, 8, 32, 128, 3, 1, 1),       # specialised wgrad kernel (C, K multiples of 64), few chunks
    (7, 128, 2, 32, 64, 3, 1, 1),       # specialised wgrad: one row pair per image, odd N
    (3, 64, 8, 64, 128, 3, 2, 1),       # specialised stride-2 kernels: non-square (4 x 32 out), odd N
    (1, 64, 4, 128, 128, 3, 2, 1),      # stride-2: two 32-column blocks per row, a single image
    (2, 128, 64, 64, 128, 3, 2, 1),     # stride-2: two big-channel tiles
    (3, 1, 32, 64, 40, 5, 1, 2),        # single-input-channel kernels: 5x5, K not a multiple of 32, non-square
    (2, 1, 16, 64, 64, 7, 2, 3),        # single-input-channel, stride 2, one row tile
    (1, 1, 24, 32, 8, 3, 1, 1),         # single-input-channel 3x3, 8 output channels, row tile of 8 with 3 tiles
    # dispatch-boundary shapes of the specialised kernels
    (1, 64, 8, 192, 128, 3, 2, 1),      # stride-2 kernels with THREE 32-column blocks per output row
    (2, 72, 8, 64, 192, 3, 2, 1),       # C % 8 == 0 but not % 64 (fwd specialised, wgrad generic), M = 192 (masked half tile)
    (1, 64, 12, 64, 128, 3, 2, 1),      # output height 6: not a multiple of 4 -> generic forward, specialised wgrad
    (2, 24, 8, 32, 192, 3, 1, 1),       # stride-1 3x3 kernel: 3 channel chunks, M = 192
    (2, 1, 32, 192, 20, 7, 1, 3),       # single-channel: Q = 192 (forward specialised, wgrad generic)
    (2, 1, 8, 128, 64, 7, 1, 3),        # single-channel: one row tile of 8, Q = 128
    (1, 128, 16, 32, 64, 3, 2, 1),      # stride 2 with 16 output columns: everything generic
    (2, 64, 16, 64, 64, 3, 2, 1),       # stride 2, M = 64: forward generic (needs M >= 128), dgrad on the BM=64 transposed kernel
    (2, 256, 32, 32, 256, 3, 1, 1),     # 3x3 kernel at a small batch: reduction split over blockIdx.z (+ reduce with bias / act)
    (1, 128, 8, 32, 128, 3, 1, 1),      # same, one image, 16 chunks
    (2, 520, 5, 7, 300, 3, 2, 1),       # generic kernels with the reduction split, ragged everything, both directions
    (3, 300, 6, 5, 520, 5, 1, 2),       # same, 5x5 stride 1
]


# LeakyReLU's derivative is discontinuous at 0: a pre-activation that rounds to +-1e-8 differently in two
# correct fp32 implementations flips one element of the gradient by 100x.  Backward is therefore checked
# tightly WITHOUT the activation (and with the smooth tanh), and with LeakyReLU at LRELU_BWD_TOL.
LRELU_BWD_TOL = 1e-2


@pytest.mark.parametrize("case", CONV_CASES, ids=lambda c: "x".join(map(str, c)))
@pytest.mark.parametrize("act", ["none", "lrelu"])
def test_conv2d_fwd_dgrad_wgrad(case, act):
    _need_gpu()
    from lsps_amd import ops
    N, C, H, W, K, R, st, pad = case
    x = _rand(N, C, H, W, seed=1).requires_grad_(True)
    w = _rand(K, C, R, R, seed=2, scale=0.1).requires_grad_(True)
    b = _rand(K, seed=3, scale=0.1).requires_grad_(True)
    y_ref = F.conv2d(x, w, b, stride=st, padding=pad)
    if act == 'lrelu':
        y_ref = F.leaky_relu(y_ref, 0.01)
    gy = _rand(*y_ref.shape, seed=4)
    y_ref.backward(gy)

    xd, wd, bd = (t.detach().cuda().requires_grad_(True) for t in (x, w, b))
    y = ops.conv2d(xd, wd, bd, st, pad, ops.ACT_LRELU if act == 'lrelu' else ops.ACT_NONE, 0.01)
    y.backward(gy.cuda())
    assert y.shape == y_ref.shape
    tol = LRELU_BWD_TOL if act == 'lrelu' else 1e-4
    errs = dict(y=_rel(y, y_ref), dx=_rel(xd.grad, x.grad), dw=_rel(wd.grad, w.grad), db=_rel(bd.grad, b.grad))
    assert errs['y'] < 1e-4 and errs['dx'] < tol and errs['dw'] < 2 * tol and errs['db'] < 2 * tol, errs


def test_stride2_kernels_seeded_random_shapes():
    """3x3 / stride 2 / pad 1 Conv2d and ConvTranspose2d (forward, dgrad, wgrad) on seeded random geometries around the
    dispatch boundaries of the specialised stride-2 kernels (1 ... 5 images, 8 ... 144 channels, 1 ... 4 column blocks, output
    heights that are / are not multiples of the row tiles): the buffer-descriptor staging (out-of-range offsets for the
    padding, wave-uniform ragged rounds, the wgrad's chunk cursor across rows and images) against torch."""
    _need_gpu()
    import random
    from lsps_amd import ops
    rng = random.Random(20260929)
    seen = set()
    for trial in range(14):
        N = rng.choice([1, 2, 3, 5])
        C = rng.choice([8, 16, 24, 64, 72, 128, 144])
        K = rng.choice([64, 128, 192, 256])
        Hs = rng.choice([4, 8, 12, 16, 20])
        Ws = rng.choice([32, 64, 96, 128])
        if N * C * 4 * Hs * Ws > 12e6:
            continue
        x = _rand(N, C, 2 * Hs, 2 * Ws, seed=100 + trial).requires_grad_(True)
        w = _rand(K, C, 3, 3, seed=200 + trial, scale=0.1).requires_grad_(True)
        b = _rand(K, seed=300 + trial, scale=0.1).requires_grad_(True)
        y = F.conv2d(x, w, b, stride=2, padding=1)
        gy = _rand(*y.shape, seed=400 + trial)
        y.backward(gy)
        xd, wd, bd = (t.detach().cuda().requires_grad_(True) for t in (x, w, b))
        ops.kernel_log_begin()
        yd = ops.conv2d(xd, wd, bd, 2, 1)
        yd.backward(gy.cuda())
        seen.update(ops.kernel_log_end())
        errs = (_rel(yd, y), _rel(xd.grad, x.grad), _rel(wd.grad, w.grad), _rel(bd.grad, b.grad))
        assert max(errs) < 2e-4, ('conv', (N, C, Hs, Ws, K), errs)
        Ci, Co = rng.choice([16, 32, 128, 144, 256]), rng.choice([64, 128, 192])
        if N * Ci * Hs * Ws > 4e6:
            continue
        xs = _rand(N, Ci, Hs, Ws, seed=500 + trial).requires_grad_(True)
        wt = _rand(Ci, Co, 3, 3, seed=600 + trial, scale=0.1).requires_grad_(True)
        yt = F.conv_transpose2d(xs, wt, None, stride=2, padding=1, output_padding=1)
        gt = _rand(*yt.shape, seed=700 + trial)
        yt.backward(gt)
        xsd, wtd = (t.detach().cuda().requires_grad_(True) for t in (xs, wt))
        ops.kernel_log_begin()
        ytd = ops.conv_transpose2d(xsd, wtd, None, 2, 1, 1)
        ytd.backward(gt.cuda())
        seen.update(ops.kernel_log_end())
        errs = (_rel(ytd, yt), _rel(xsd.grad, xs.grad), _rel(wtd.grad, wt.grad))
        assert max(errs) < 2e-4, ('convT', (N, Ci, Hs, Ws, Co), errs)
    assert {'igemm_f3x3s2_kernel', 'igemm_t3x3s2_kernel', 'igemm_w3x3s2_kernel'} <= seen, seen


# (N, C, H, W, K, R, stride, pad, groups)
GROUPED_CASES = [
    (2, 512, 32, 32, 512, 3, 1, 1, 8),   # LeakyINSResNeXtBlock at full width: k * inplanes = 512, cardinality 8 (common_net.py:116)
    (3, 16, 32, 32, 16, 3, 1, 1, 8),     # the tiny-width golden geometry (2 channels per group)
    (2, 12, 9, 7, 18, 3, 2, 1, 3),       # ragged, stride 2, C/G != K/G
    (1, 96, 8, 8, 192, 1, 1, 0, 2),      # 1x1 grouped, K/G = 96
    (2, 8, 6, 6, 8, 3, 1, 1, 1),         # groups = 1 falls through to the dense op
]


@pytest.mark.parametrize("case", GROUPED_CASES, ids=lambda c: "x".join(map(str, c)))
@pytest.mark.parametrize("bias", [True, False])
def test_conv2d_grouped_fwd_dgrad_wgrad(case, bias):
    """lsps_conv2d_grouped_{fwd,dgrad,wgrad} (channel-slice launches, no copies) against F.conv2d(groups=G)."""
    _need_gpu()
    from lsps_amd import ops
    N, C, H, W, K, R, st, pad, G = case
    x = _rand(N, C, H, W, seed=1).requires_grad_(True)
    w = _rand(K, C // G, R, R, seed=2, scale=0.1).requires_grad_(True)
    b = _rand(K, seed=3, scale=0.1).requires_grad_(True) if bias else None
    y_ref = F.conv2d(x, w, b, stride=st, padding=pad, groups=G)
    gy = _rand(*y_ref.shape, seed=4)
    y_ref.backward(gy)
    xd, wd = (t.detach().cuda().requires_grad_(True) for t in (x, w))
    bd = b.detach().cuda().requires_grad_(True) if bias else None
    y = ops.conv2d_grouped(xd, wd, bd, st, pad, G)
    y.backward(gy.cuda())
    assert y.shape == y_ref.shape
    errs = dict(y=_rel(y, y_ref), dx=_rel(xd.grad, x.grad), dw=_rel(wd.grad, w.grad))
    if bias:
        errs['db'] = _rel(bd.grad, b.grad)
    assert all(e < 2e-4 for e in errs.values()), errs


# (N, Ci, H, W, Co, R, stride, pad, outpad)
CONVT_CASES = [
    (2, 256, 32, 32, 128, 3, 2, 1, 1),   # gen up 1
    (2, 128, 64, 64, 64, 3, 2, 1, 1),    # gen up 2
    (2, 64, 128, 128, 1, 1, 1, 0, 0),    # gen output 1x1
    (3, 20, 1, 1, 64, 4, 1, 0, 0),       # Mapping layer 0 (narrow)
    (3, 64, 4, 4, 48, 4, 2, 1, 0),       # Mapping 4x4 stride 2
    (2, 7, 5, 6, 9, 3, 2, 1, 1),         # ragged
    (2, 6, 7, 5, 10, 3, 1, 1, 0),        # stride 1
    (2, 5, 6, 6, 4, 5, 3, 2, 2),         # stride 3
    (1, 128, 2, 32, 64, 3, 2, 1, 1),     # specialised stride-2 kernels: two input rows, a single image
    (3, 128, 4, 64, 128, 3, 2, 1, 1),    # stride-2: two 32-column blocks, odd N
    (1, 144, 8, 96, 64, 3, 2, 1, 1),     # Ci = 144 (9 chunks of 16), three column blocks, 64 outputs (BM = 64, TR = 8)
    (1, 144, 4, 96, 64, 3, 2, 1, 1),     # H = 4 with 64 outputs: TR = 8 does not divide -> generic transposed path
    (2, 40, 8, 32, 192, 3, 2, 1, 1),     # Ci % 16 != 0 -> generic forward; dgrad specialised (C % 8 == 0), Co = 192
    (2, 64, 6, 32, 128, 3, 2, 1, 1),     # H = 6: not a multiple of 4 -> generic
]


@pytest.mark.parametrize("case", CONVT_CASES, ids=lambda c: "x".join(map(str, c)))
@pytest.mark.parametrize("act", ["lrelu", "tanh", "none"])
def test_conv_transpose2d(case, act):
    _need_gpu()
    from lsps_amd import ops
    N, Ci, H, W, Co, R, st, pad, op = case
    x = _rand(N, Ci, H, W, seed=5).requires_grad_(True)
    w = _rand(Ci, Co, R, R, seed=6, scale=0.1).requires_grad_(True)
    b = _rand(Co, seed=7, scale=0.1).requires_grad_(True)
    pre = F.conv_transpose2d(x, w, b, stride=st, padding=pad, output_padding=op)
    y_ref = {'lrelu': lambda t: F.leaky_relu(t, 0.01), 'tanh': torch.tanh, 'none': lambda t: t}[act](pre)
    gy = _rand(*y_ref.shape, seed=8)
    y_ref.backward(gy)
    code = {'lrelu': ops.ACT_LRELU, 'tanh': ops.ACT_TANH, 'none': ops.ACT_NONE}[act]
    xd, wd, bd = (t.detach().cuda().requires_grad_(True) for t in (x, w, b))
    y = ops.conv_transpose2d(xd, wd, bd, st, pad, op, code, 0.01)
    y.backward(gy.cuda())
    assert y.shape == y_ref.shape
    tol = LRELU_BWD_TOL if act == 'lrelu' else 1e-4
    errs = dict(y=_rel(y, y_ref), dx=_rel(xd.grad, x.grad), dw=_rel(wd.grad, w.grad), db=_rel(bd.grad, b.grad))
    assert errs['y'] < 1e-4 and errs['dx'] < tol and errs['dw'] < 2 * tol and errs['db'] < 2 * tol, errs


@pytest.mark.parametrize("shape", [(3, 256, 32, 32), (2, 8, 64, 64), (2, 5, 7, 9), (1, 3, 128, 128)])
@pytest.mark.parametrize("variant", ["lrelu", "residual", "plain"])
def test_instance_norm_fused(shape, variant):
    _need_gpu()
    from lsps_amd import ops
    y = _rand(*shape, seed=9, scale=3.0).requires_grad_(True)
    r = _rand(*shape, seed=10).requires_grad_(True)
    ref = F.instance_norm(y, eps=1e-5)
    if variant == 'lrelu':
        ref = F.leaky_relu(ref, 0.01)
    elif variant == 'residual':
        ref = ref + r
    g = _rand(*shape, seed=11)
    ref.backward(g)

    yd = y.detach().cuda().requires_grad_(True)
    rd = r.detach().cuda().requires_grad_(True)
    pre = yd * 1.0                                   # a non-leaf the op may overwrite in place
    out = ops.instance_norm_(pre, rd if variant == 'residual' else None, 0.01 if variant == 'lrelu' else -1.0)
    out.backward(g.cuda())
    assert _rel(out, ref) < 1e-4
    assert _rel(yd.grad, y.grad) < (LRELU_BWD_TOL if variant == 'lrelu' else 2e-4)
    if variant == 'residual':
        assert _rel(rd.grad, r.grad) < 1e-6


@pytest.mark.parametrize("n", [1, 7, 4096, 100003, 3 * 1024 * 1024 + 5])
def test_losses(n):
    _need_gpu()
    from lsps_amd import ops
    a = _rand(n, seed=12).requires_grad_(True)
    b = _rand(n, seed=13).requires_grad_(True)
    ad, bd = a.detach().cuda().requires_grad_(True), b.detach().cuda().requires_grad_(True)
    cases = [
        (lambda p, q: (p - q).abs().mean(), lambda p, q: ops.l1_loss(p, q)),
        (lambda p, q: ((p - q) ** 2).mean(), lambda p, q: ops.l2_loss(p, q)),
        (lambda p, q: (p ** 2).mean() + 0 * q.sum(), lambda p, q: ops.kl_loss(p)),
    ]
    for ref_fn, hip_fn in cases:
        for t in (a, b, ad, bd):
            t.grad = None
        lr = ref_fn(a, b) * 3.0
        lr.backward()
        lh = hip_fn(ad, bd) * 3.0
        lh.backward()
        assert abs(lh.item() - lr.item()) <= 1e-5 * max(1.0, abs(lr.item()))
        assert _rel(ad.grad, a.grad) < 1e-5
        if bd.grad is not None and b.grad is not None and b.grad.abs().max() > 0:
            assert _rel(bd.grad, b.grad) < 1e-5


def test_l1_against_zero_and_kl_sd():
    _need_gpu()
    from lsps_amd import ops
    a = _rand(6, 20, seed=14).requires_grad_(True)
    sd = (_rand(6, 20, seed=15).abs() + 0.1).requires_grad_(True)
    ad, sdd = a.detach().cuda().requires_grad_(True), sd.detach().cuda().requires_grad_(True)
    ref = (a ** 2 + sd ** 2 - torch.log(sd ** 2)).sum() / a.size(0)
    ref.backward()
    hip = ops.kl_loss(ad, sdd)
    hip.backward()
    assert abs(hip.item() - ref.item()) < 1e-5 * abs(ref.item())
    assert _rel(ad.grad, a.grad) < 1e-5 and _rel(sdd.grad, sd.grad) < 1e-5
    z = ops.l1_loss(ad.detach())
    assert abs(z.item() - a.detach().abs().mean().item()) < 1e-6


@pytest.mark.parametrize("n", [8, 513, 4 * 768])
@pytest.mark.parametrize("target", [1.0, 0.0])
def test_bce_sigmoid(n, target):
    _need_gpu()
    from lsps_amd import ops
    x = (_rand(n, seed=16) * 6).requires_grad_(True)
    with torch.no_grad():
        x[0] = 40.0       # saturates sigmoid in fp32: exercises torch's log clamp at -100
        x[1] = -40.0
    p = torch.sigmoid(x)
    ref = F.binary_cross_entropy(p, torch.full_like(p, target))
    ref.backward()
    xd = x.detach().cuda().requires_grad_(True)
    loss, counts = ops.bce_sigmoid(xd, target)
    loss.backward()
    assert abs(loss.item() - ref.item()) <= 1e-5 * max(1.0, abs(ref.item()))
    assert _rel(xd.grad, x.grad) < 1e-5
    assert int(counts[0].item()) == int((p >= 0.5).sum().item())
    assert int(counts[1].item()) == int((p <= 0.5).sum().item())


@pytest.mark.parametrize("act", ["none", "lrelu", "softplus"])
def test_linear(act):
    _need_gpu()
    from lsps_amd import ops
    x = _rand(9, 108, seed=17).requires_grad_(True)
    w = _rand(50, 108, seed=18, scale=0.2).requires_grad_(True)
    b = _rand(50, seed=19, scale=0.2).requires_grad_(True)
    pre = F.linear(x, w, b)
    ref = {'none': lambda t: t, 'lrelu': lambda t: F.leaky_relu(t, 0.01), 'softplus': F.softplus}[act](pre)
    g = _rand(9, 50, seed=20)
    ref.backward(g)
    code = {'none': ops.ACT_NONE, 'lrelu': ops.ACT_LRELU, 'softplus': ops.ACT_SOFTPLUS}[act]
    xd, wd, bd = (t.detach().cuda().requires_grad_(True) for t in (x, w, b))
    y = ops.linear(xd, wd, bd, code, 0.01)
    y.backward(g.cuda())
    assert _rel(y, ref) < 1e-5
    assert _rel(xd.grad, x.grad) < 1e-5 and _rel(wd.grad, w.grad) < 1e-5 and _rel(bd.grad, b.grad) < 1e-5


def test_flat_adam_matches_torch_adam():
    """FlatAdam over the arena == torch.optim.Adam, including the skip of gradient-less tensors and
    per-parameter step counts."""
    _need_gpu()
    from lsps_amd.optim import FlatAdam
    shapes = [(64, 1, 7, 7), (64,), (5000,), (3, 3), (20, 2048, 2, 2)]
    ps_ref = [torch.nn.Parameter(_rand(*s, seed=30 + i, scale=0.05)) for i, s in enumerate(shapes)]
    ps_hip = [torch.nn.Parameter(p.detach().clone().cuda()) for p in ps_ref]
    ref = torch.optim.Adam(ps_ref, lr=1e-3, betas=(0.5, 0.999), weight_decay=1e-4)
    hip = FlatAdam(ps_hip, lr=1e-3, betas=(0.5, 0.999), weight_decay=1e-4)
    hip.attach()
    for step in range(4):
        ref.zero_grad()
        hip.zero_grad()
        for i, (pr, ph) in enumerate(zip(ps_ref, ps_hip)):
            if step % 2 == 1 and i == 2:
                continue                      # tensor 2 gets no gradient on odd steps
            g = _rand(*pr.shape, seed=100 * step + i, scale=0.01)
            (pr * g).sum().backward()
            (ph * g.cuda()).sum().backward()
        ref.step()
        hip.step()
    for pr, ph in zip(ps_ref, ps_hip):
        assert _rel(ph, pr) < 1e-5


def test_flat_adam_state_dict_round_trip():
    """Optimizer state saved by torch.optim.Adam loads into FlatAdam (resume(load_opt=True)) and continues identically."""
    _need_gpu()
    from lsps_amd.optim import FlatAdam
    shapes = [(16, 3, 3, 3), (16,), (100,)]
    ps_ref = [torch.nn.Parameter(_rand(*s, seed=40 + i, scale=0.05).cuda()) for i, s in enumerate(shapes)]
    ps_hip = [torch.nn.Parameter(p.detach().clone()) for p in ps_ref]
    ref = torch.optim.Adam(ps_ref, lr=1e-3, betas=(0.5, 0.999), weight_decay=1e-4)
    hip = FlatAdam(ps_hip, lr=1e-3, betas=(0.5, 0.999), weight_decay=1e-4)
    hip.attach()

    def grads(step):
        for pr, ph in zip(ps_ref, ps_hip):
            g = _rand(*pr.shape, seed=200 + step, scale=0.01).cuda()
            pr.grad = g.clone()
            ph.grad.copy_(g)
        hip.arena.touched = [True] * len(ps_hip)
    for step in range(2):
        grads(step)
        ref.step()
        if step == 0:
            hip.step()
    # hip is one step behind: load ref's state (after 2 steps) and parameters, then both take step 3
    hip.load_state_dict(ref.state_dict())
    for pr, ph in zip(ps_ref, ps_hip):
        ph.data.copy_(pr.data)
    grads(2)
    ref.step()
    hip.step()
    for pr, ph in zip(ps_ref, ps_hip):
        assert _rel(ph, pr) < 1e-6


def test_ops_reject_cpu_tensors():
    from lsps_amd import ops, _lib
    with pytest.raises(Exception):
        ops.conv2d(torch.zeros(1, 1, 4, 4), torch.zeros(1, 1, 3, 3), None, 1, 1)


@pytest.mark.parametrize("case", [(3, 256, 32, 32, 256, 3, 1, 1), (2, 8, 32, 32, 130, 3, 1, 1), (5, 16, 8, 32, 128, 3, 1, 1),
                                  (3, 64, 8, 32, 128, 3, 1, 1), (7, 128, 2, 32, 64, 3, 1, 1),
                                  (2, 64, 128, 128, 128, 3, 2, 1), (4, 512, 8, 8, 1024, 3, 2, 1), (2, 1, 128, 128, 64, 7, 1, 3),
                                  (5, 3, 17, 13, 37, 3, 1, 1), (6, 2048, 2, 2, 20, 2, 1, 0),
                                  (2, 128, 64, 64, 256, 3, 2, 1), (1, 64, 8, 192, 128, 3, 2, 1), (3, 144, 16, 64, 64, 3, 2, 1)],
                         ids=lambda c: "x".join(map(str, c)))
def test_conv3x3_bf16_math_mode(case):
    """BASELINE config 5: the MFMA conv kernels (specialised 3x3 and generic) with bf16 operands (f32 accumulate, f32 tensors).
    Tolerance 1e-2 of the abs-max: operands carry 8 mantissa bits (2^-9 relative rounding each)."""
    _need_gpu()
    from lsps_amd import ops
    N, C, H, W, K, R, st, pad = case
    x = _rand(N, C, H, W, seed=1).requires_grad_(True)
    w = _rand(K, C, R, R, seed=2, scale=0.1).requires_grad_(True)
    y_ref = F.conv2d(x, w, None, stride=st, padding=pad)
    gy = _rand(*y_ref.shape, seed=4)
    y_ref.backward(gy)
    xd, wd = (t.detach().cuda().requires_grad_(True) for t in (x, w))
    assert ops.get_math_mode() == 'f32'
    ops.set_math_mode('bf16')
    try:
        y = ops.conv2d(xd, wd, None, st, pad)
        y.backward(gy.cuda())
        torch.cuda.synchronize()
    finally:
        ops.set_math_mode('f32')
    errs = dict(y=_rel(y, y_ref), dx=_rel(xd.grad, x.grad), dw=_rel(wd.grad, w.grad))
    assert all(e < 1e-2 for e in errs.values()), errs
    assert max(errs.values()) > 1e-5, "bf16 mode did not engage (result is f32-exact)"


@pytest.mark.parametrize("case", [(3, 256, 32, 32, 256, 3, 1, 1), (2, 8, 32, 32, 130, 3, 1, 1), (3, 64, 8, 32, 128, 3, 1, 1)],
                         ids=lambda c: "x".join(map(str, c)))
def test_conv3x3_f32_split_math_mode(case):
    """Experimental mode 2: f32 products from three bf16 limbs per operand (x = hi + mid + lo exactly) and six
    bf16 MFMAs (dropped cross terms < 2^-24 relative).  Same tolerance as the exact-f32 kernels."""
    _need_gpu()
    from lsps_amd import ops
    N, C, H, W, K, R, st, pad = case
    x = _rand(N, C, H, W, seed=1).requires_grad_(True)
    w = _rand(K, C, R, R, seed=2, scale=0.1).requires_grad_(True)
    y_ref = F.conv2d(x, w, None, stride=st, padding=pad)
    gy = _rand(*y_ref.shape, seed=4)
    y_ref.backward(gy)
    xd, wd = (t.detach().cuda().requires_grad_(True) for t in (x, w))
    ops.set_math_mode('f32_split')
    try:
        y = ops.conv2d(xd, wd, None, st, pad)
        y.backward(gy.cuda())
        torch.cuda.synchronize()
    finally:
        ops.set_math_mode('f32')
    errs = dict(y=_rel(y, y_ref), dx=_rel(xd.grad, x.grad), dw=_rel(wd.grad, w.grad))
    assert errs['y'] < 1e-4 and errs['dx'] < 1e-4 and errs['dw'] < 2e-4, errs


@pytest.mark.parametrize("case", [(2, 256, 8, 32, 256, 3, 1, 1),      # fused: addend in the 3x3 kernel's epilogue
                                  (3, 12, 9, 11, 20, 5, 2, 2),        # fallback: dgrad, then an add pass
                                  (2, 1, 32, 32, 16, 7, 1, 3)],       # fallback through the single-channel tap-GEMM dgrad
                         ids=lambda c: "x".join(map(str, c)))
def test_conv2d_dgrad_acc(case):
    """lsps_conv2d_dgrad_acc: dx = dgrad(dy, w) + addend (the skip-connection gradient of a residual block)."""
    _need_gpu()
    from lsps_amd import _lib
    N, C, H, W, K, R, st, pad = case
    L = _lib.lib()
    x = _rand(N, C, H, W, seed=1).requires_grad_(True)
    w = _rand(K, C, R, R, seed=2, scale=0.1)
    y = F.conv2d(x, w, None, stride=st, padding=pad)
    gy = _rand(*y.shape, seed=4)
    y.backward(gy)
    add = _rand(N, C, H, W, seed=9)
    want = x.grad + add
    gyd, wd, addd = gy.cuda().contiguous(), w.cuda().contiguous(), add.cuda().contiguous()
    dx = torch.empty(N, C, H, W, device='cuda')
    ws, wsb = _lib.workspace(L.lsps_conv2d_workspace_bytes(N, C, H, W, K, R, R, st, pad), dx.device)
    _lib.check(L.lsps_conv2d_dgrad_acc(_lib.ptr(gyd), _lib.ptr(wd), _lib.ptr(addd), _lib.ptr(dx), N, C, H, W, K, R, R, st,
                                       pad, ws, wsb, _lib.stream()), 'dgrad_acc')
    assert _rel(dx, want) < 1e-4


def test_empty_batch_behaves_like_torch():
    """Zero samples in -> zero samples out with the right shape; parameter gradients of an empty batch are zeros (what
    torch.nn returns); no kernel is launched."""
    _need_gpu()
    from lsps_amd import ops
    import lsps_amd.trainers as tr
    dev = 'cuda'
    x0 = torch.zeros(0, 8, 32, 32, device=dev, requires_grad=True)
    w = torch.randn(16, 8, 3, 3, device=dev, requires_grad=True)
    b = torch.randn(16, device=dev, requires_grad=True)
    y = ops.conv2d(x0, w, b, 2, 1, ops.ACT_LRELU, 0.01)
    assert tuple(y.shape) == (0, 16, 16, 16)
    y.sum().backward()
    assert tuple(x0.grad.shape) == (0, 8, 32, 32) and float(w.grad.abs().max()) == 0 and float(b.grad.abs().max()) == 0
    wt = torch.randn(8, 4, 3, 3, device=dev)
    assert tuple(ops.conv_transpose2d(x0.detach(), wt, None, 2, 1, 1).shape) == (0, 4, 64, 64)
    assert tuple(ops.instance_norm_(torch.zeros(0, 8, 4, 4, device=dev), None, 0.01).shape) == (0, 8, 4, 4)
    assert tuple(ops.linear(torch.zeros(0, 12, device=dev), torch.randn(5, 12, device=dev), torch.randn(5, device=dev)).shape) == (0, 5)
    blk = tr.LeakyINSResBlock(128, 128).cuda()
    assert tuple(blk(torch.zeros(0, 128, 32, 32, device=dev)).shape) == (0, 128, 32, 32)
    hp = __import__('yaml').safe_load(open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))),
                                                        'exps', 'nnyu.yaml')))['train']['hyperparameters']
    from lsps_amd import synth
    hp = synth.tiny_hyperparameters(hp)
    gen, dis = tr.SharedResGen(hp['gen']).cuda().eval(), tr.SharedDis(hp['dis']).cuda().eval()
    e = torch.zeros(0, 1, 128, 128, device=dev)
    with torch.no_grad():
        za, zb = gen.encode(e, e)               # (gen.forward itself splits by x_A.size(0) == 0: an error in the reference too)
        assert za.shape[0] == 0 and zb.shape[0] == 0
        oa, ob = gen.decode(za)
        assert tuple(oa.shape) == (0, 1, 128, 128) and tuple(ob.shape) == (0, 1, 128, 128)
        assert dis.model_S(dis.model_A(e)).shape[0] == 0


# (N, C, H, K): Winograd F(2x2,3x3) path of the 3x3 / stride-1 / width-32 convs (forced for every eligible shape)
WINO_CASES = [
    (3, 256, 32, 256),     # the residual conv; odd N * 4 tiles: 12 pixel tiles, 4 k slices (XCD mapping)
    (1, 32, 8, 64),        # one pixel tile, one k slice, two row chunks: top AND bottom halo rows in one workgroup
    (5, 64, 8, 128),       # odd pixel-tile count: plain workgroup order
    (2, 96, 16, 192),      # 3 k slices, 6 row-chunk pairs
    (2, 48, 24, 64),       # H = 24: three tiles per image
    (2, 64, 4, 64),        # wgrad: two tile rows per image (first AND last), one 64x64 block
    (3, 128, 6, 64),       # wgrad: 9 tile rows (odd chunk count per split)
    (1, 64, 32, 128),      # wgrad: a single image over 16 splits
    (17, 32, 32, 256),     # 272 workgroups of 64 channels: the 512-thread forward kernel, two-slices-per-XCD mapping
    (33, 16, 16, 256),     # 264 workgroups, pixel-tile count 66 (not a multiple of 4): one-slice-per-XCD mapping
]


@pytest.mark.parametrize("case", WINO_CASES, ids=lambda c: "x".join(map(str, c)))
@pytest.mark.parametrize("act", ["none", "lrelu"])
def test_conv3x3_winograd(case, act):
    """Winograd kernel vs an f64 convolution: forward (bias, activation), dgrad (flipped taps) and the fused
    dgrad + addend; same tolerance class as the direct kernel (both are f32 MFMA accumulations)."""
    _need_gpu()
    from lsps_amd import _lib, ops
    N, C, H, K = case
    L = _lib.lib()
    prev = ops.get_winograd()
    ops.set_winograd('always_f2')
    try:
        x = _rand(N, C, H, 32, seed=1).double().requires_grad_(True)
        w = _rand(K, C, 3, 3, seed=2, scale=0.1).double().requires_grad_(True)
        b = _rand(K, seed=3, scale=0.1).double().requires_grad_(True)
        y_ref = F.conv2d(x, w, b, padding=1)
        if act == 'lrelu':
            y_ref = F.leaky_relu(y_ref, 0.01)
        gy = _rand(*y_ref.shape, seed=4)
        y_ref.backward(gy.double())
        xd, wd, bd = (t.detach().float().cuda().requires_grad_(True) for t in (x, w, b))
        y = ops.conv2d(xd, wd, bd, 1, 1, ops.ACT_LRELU if act == 'lrelu' else ops.ACT_NONE, 0.01)
        y.backward(gy.cuda())
        errs = dict(y=_rel(y, y_ref), dx=_rel(xd.grad, x.grad), dw=_rel(wd.grad, w.grad), db=_rel(bd.grad, b.grad))
        tol = LRELU_BWD_TOL if act == 'lrelu' else 2e-5
        assert errs['y'] < 2e-5 and errs['dx'] < tol and errs['dw'] < 1e-4 and errs['db'] < 1e-4, errs
        # against the direct kernel: round-off only
        ops.set_winograd('off')
        y_dir = ops.conv2d(xd.detach(), wd.detach(), bd.detach(), 1, 1, ops.ACT_LRELU if act == 'lrelu' else ops.ACT_NONE, 0.01)
        assert _rel(y, y_dir) < 2e-5
        ops.set_winograd('always_f2')
        if act == 'none' and C % 64 == 0 and K % 32 == 0:
            # fused dgrad + addend (residual block backward): dgrad's "input channels" are K, its outputs C
            add = _rand(N, C, H, 32, seed=9)
            gyd, wdd, addd = gy.cuda().contiguous(), wd.detach().contiguous(), add.cuda().contiguous()
            dx = torch.empty(N, C, H, 32, device='cuda')
            ws, wsb = _lib.workspace(L.lsps_conv2d_workspace_bytes(N, C, H, 32, K, 3, 3, 1, 1), dx.device)
            _lib.check(L.lsps_conv2d_dgrad_acc(_lib.ptr(gyd), _lib.ptr(wdd), _lib.ptr(addd), _lib.ptr(dx), N, C, H, 32, K,
                                               3, 3, 1, 1, ws, wsb, _lib.stream()), 'dgrad_acc')
            assert _rel(dx, x.grad.float() + add) < 2e-5
    finally:
        ops.set_winograd(prev)


def test_winograd_mode_switch():
    _need_gpu()
    from lsps_amd import _lib, ops
    prev = ops.get_winograd()
    try:
        for m in ('off', 'always', 'auto_f2', 'always_f2', 'auto'):
            ops.set_winograd(m)
            assert ops.get_winograd() == m
        assert _lib.lib().lsps_set_winograd(5) != 0
    finally:
        ops.set_winograd(prev)


def test_winograd_inside_weight_cache_scope():
    """The transformed weights are cached with the other packed panels: two forwards and a dgrad inside one scope give
    the same results as outside it, and a weight update between scopes is seen."""
    _need_gpu()
    from lsps_amd import ops
    prev = ops.get_winograd()
    ops.set_winograd('always')
    try:
        x = _rand(2, 64, 16, 32, seed=1).cuda()
        w = _rand(64, 64, 3, 3, seed=2, scale=0.1).cuda()
        y0 = ops.conv2d(x, w, None, 1, 1)
        ops.weight_cache_begin(x.device)
        try:
            y1 = ops.conv2d(x, w, None, 1, 1)
            y2 = ops.conv2d(x, w, None, 1, 1)
        finally:
            ops.weight_cache_end()
        assert torch.equal(y0, y1) and torch.equal(y0, y2)
        w.mul_(2.0)
        ops.weight_cache_begin(x.device)
        try:
            y3 = ops.conv2d(x, w, None, 1, 1)
        finally:
            ops.weight_cache_end()
        assert _rel(y3, 2.0 * y0) < 1e-6
    finally:
        ops.set_winograd(prev)


def test_conv3x3_winograd_random_shapes_against_direct():
    """Seeded random eligible shapes: Winograd forward, dgrad and weight gradient against the direct kernels
    (both through the C-ABI; round-off apart they compute the same thing)."""
    _need_gpu()
    from lsps_amd import _lib, ops
    L = _lib.lib()
    st = _lib.stream()
    rng = np.random.RandomState(1234)
    prev = ops.get_winograd()
    try:
        for it in range(16):
            N = int(rng.randint(1, 12))
            C = 64 * int(rng.randint(1, 4))
            K = 64 * int(rng.randint(1, 4))
            H = 8 * int(rng.randint(1, 5))
            x = _rand(N, C, H, 32, seed=100 + it).cuda()
            w = _rand(K, C, 3, 3, seed=200 + it, scale=0.1).cuda()
            b = _rand(K, seed=300 + it, scale=0.1).cuda()
            gy = _rand(N, K, H, 32, seed=400 + it).cuda()
            ws, wsb = _lib.workspace(L.lsps_conv2d_workspace_bytes(N, C, H, 32, K, 3, 3, 1, 1), x.device)
            res = {}
            for mode in ('off', 'always_f2'):
                ops.set_winograd(mode)
                y = torch.empty(N, K, H, 32, device='cuda')
                dx = torch.empty_like(x)
                dw = torch.empty_like(w)
                _lib.check(L.lsps_conv2d_fwd(_lib.ptr(x), _lib.ptr(w), _lib.ptr(b), _lib.ptr(y), N, C, H, 32, K, 3, 3, 1, 1,
                                             ops.ACT_LRELU, 0.2, ws, wsb, st), 'fwd')
                _lib.check(L.lsps_conv2d_dgrad(_lib.ptr(gy), _lib.ptr(w), _lib.ptr(dx), N, C, H, 32, K, 3, 3, 1, 1, ws, wsb,
                                               st), 'dgrad')
                _lib.check(L.lsps_conv2d_wgrad(_lib.ptr(x), _lib.ptr(gy), _lib.ptr(dw), None, N, C, H, 32, K, 3, 3, 1, 1, ws,
                                               wsb, st), 'wgrad')
                res[mode] = (y, dx, dw)
            for a, d, name in zip(res['always_f2'], res['off'], ('y', 'dx', 'dw')):
                assert _rel(a, d) < 3e-5, (it, N, C, K, H, name, _rel(a, d))
    finally:
        ops.set_winograd(prev)


# (N, C, K) on 32x32 maps: Winograd F(4x4,3x3) (conv_wino4.h).  Its f32 round-off is ~1e-5 of the output's abs-max (the
# transforms multiply by up to 8 / divide by 24), an order above the direct and F(2x2,3x3) kernels and two orders inside
# north_star's 1e-3: it gets its own bound, 5e-5 against an f64 convolution.
WINO4_CASES = [
    (3, 256, 256),      # the residual conv; odd N: plain workgroup order
    (4, 256, 256),      # even N, 8 k slices: two-slices-per-XCD mapping
    (1, 8, 32),         # one workgroup, a single 8-channel row chunk
    (2, 24, 96),        # odd chunk count (3), 3 k slices
    (5, 128, 64),
    (18, 16, 256),      # 144 workgroups: 'auto' territory
]
W4_TOL = 5e-5


@pytest.mark.parametrize("case", WINO4_CASES, ids=lambda c: "x".join(map(str, c)))
def test_conv3x3_winograd_f4(case):
    """F(4x4,3x3) forward (bias + LeakyReLU epilogue), dgrad (flipped taps, C and K swap roles) and dgrad + addend against an
    f64 convolution, and the kernel the library reports."""
    _need_gpu()
    from lsps_amd import _lib, ops
    N, C, K = case
    L = _lib.lib()
    prev = ops.get_winograd()
    ops.set_winograd('always')
    try:
        x = _rand(N, C, 32, 32, seed=11).double().requires_grad_(True)
        w = _rand(K, C, 3, 3, seed=12, scale=0.1).double().requires_grad_(True)
        b = _rand(K, seed=13, scale=0.1).double()
        y_ref = F.leaky_relu(F.conv2d(x, w, b, padding=1), 0.01)
        pre = F.conv2d(x, w, None, padding=1)
        gy = _rand(N, K, 32, 32, seed=14)
        pre.backward(gy.double())
        xd, wd, bd = x.detach().float().cuda(), w.detach().float().cuda(), b.float().cuda()
        ops.kernel_log_begin()
        y = ops.conv2d(xd, wd, bd, 1, 1, ops.ACT_LRELU, 0.01)
        names = ops.kernel_log_end()
        assert names == ['wino4_f3x3_kernel'], names
        assert _rel(y, y_ref) < W4_TOL, _rel(y, y_ref)
        ws, wsb = _lib.workspace(L.lsps_conv2d_workspace_bytes(N, C, 32, 32, K, 3, 3, 1, 1), xd.device)
        gyd = gy.cuda().contiguous()
        dx = torch.empty_like(xd)
        _lib.check(L.lsps_conv2d_dgrad(_lib.ptr(gyd), _lib.ptr(wd), _lib.ptr(dx), N, C, 32, 32, K, 3, 3, 1, 1, ws, wsb,
                                       _lib.stream()), 'dgrad')
        if C % 32 == 0 and K % 8 == 0:
            assert L.lsps_last_kernel(None) == b'wino4_f3x3_kernel'
            assert _rel(dx, x.grad) < W4_TOL, _rel(dx, x.grad)
            add = _rand(N, C, 32, 32, seed=19).cuda()
            _lib.check(L.lsps_conv2d_dgrad_acc(_lib.ptr(gyd), _lib.ptr(wd), _lib.ptr(add), _lib.ptr(dx), N, C, 32, 32, K, 3, 3,
                                               1, 1, ws, wsb, _lib.stream()), 'dgrad_acc')
            assert _rel(dx, x.grad.float() + add.cpu()) < W4_TOL
        else:
            assert _rel(dx, x.grad) < 2e-5
    finally:
        ops.set_winograd(prev)


@pytest.mark.parametrize("case", [(3, 256, 32, 256), (4, 64, 32, 32), (2, 32, 16, 64), (1, 24, 8, 16)],
                         ids=lambda c: "x".join(map(str, c)))
def test_conv3x3_dgrad_through_instance_norm_fused(case):
    """lsps_conv2d_dgrad_inbwd = conv dgrad pushed through the InstanceNorm + LeakyReLU in front of the conv
    (common_net.py:168-171 <- :162 in backward) against f64 autograd of the composition: the F(4x4,3x3) epilogue path on
    32x32 maps, dgrad + in-place norm backward elsewhere.  (N, C, H, K): the conv maps C -> K channels, dx has C."""
    _need_gpu()
    from lsps_amd import _lib, ops
    N, C, H, K = case
    L = _lib.lib()
    prev = ops.get_winograd()
    ops.set_winograd('always')
    try:
        c0 = _rand(N, C, H, 32, seed=41).double().requires_grad_(True)          # pre-norm tensor
        w = _rand(K, C, 3, 3, seed=42, scale=0.1)
        gy = _rand(N, K, H, 32, seed=43)
        a1 = F.leaky_relu(F.instance_norm(c0, eps=1e-5), 0.01)
        F.conv2d(a1, w.double(), None, padding=1).backward(gy.double())
        rstd = 1.0 / torch.sqrt(c0.detach().var(dim=(2, 3), unbiased=False) + 1e-5)
        a1d, rd = a1.detach().float().cuda().contiguous(), rstd.float().reshape(-1).cuda().contiguous()
        dx = torch.full((N, C, H, 32), float('nan'), device='cuda')
        ws, wsb = _lib.workspace(L.lsps_conv2d_workspace_bytes(N, C, H, 32, K, 3, 3, 1, 1), dx.device)
        gyd, wd = gy.cuda(), w.cuda()                     # named: a temporary's block could be reused within the call expression
        _lib.check(L.lsps_conv2d_dgrad_inbwd(_lib.ptr(gyd), _lib.ptr(wd), _lib.ptr(a1d), _lib.ptr(rd), _lib.ptr(dx),
                                             N, C, H, 32, K, 0.01, ws, wsb, _lib.stream()), 'dgrad_inbwd')
        f4 = H == 32 and C % 32 == 0 and K % 8 == 0
        assert (L.lsps_last_kernel(None) == b'wino4_f3x3_kernel') == f4, L.lsps_last_kernel(None)
        # where the normalised value is within round-off of 0 the LeakyReLU slope is decided by that round-off (see above)
        xhat = F.instance_norm(c0.detach(), eps=1e-5)
        diff = (dx.cpu().double() - c0.grad).abs() / c0.grad.abs().max()
        off = diff > 1e-4
        assert int(off.sum()) <= 64 and bool((xhat[off].abs() < 1e-3).all()), (int(off.sum()), float(diff.max()))
    finally:
        ops.set_winograd(prev)


# (N, C, H, K): weight gradient in F(4x4,3x3) form (conv_wino4w.h): 64 k x 32 c blocks, tile rows of 4 image rows split over
# 256 / blocks workgroups (a multiple of 8 where possible), partial sums reduced in double precision
WINO4W_CASES = [
    (3, 256, 32, 256),     # the residual conv: 32 blocks x 8 splits of 3 tile rows (XCD mapping)
    (17, 256, 32, 256),    # 136 tile rows over 8 splits: odd chunk counts per split
    (1, 64, 4, 64),        # ONE tile row (top and bottom halo rows in the same chunk), one split
    (2, 64, 8, 128),       # 4 tile rows, 4 blocks -> 4 splits of one chunk each
    (5, 128, 12, 64),      # H = 12: 15 tile rows over 8 splits of 2: the last split is short, none empty
    (1, 64, 20, 192),      # 5 tile rows, 6 blocks -> 5 splits (not a multiple of 8: plain workgroup order)
    (9, 192, 32, 64),      # 72 tile rows, 6 blocks -> 40 splits of 2: the last four splits are EMPTY (zero partials)
]


@pytest.mark.parametrize("case", WINO4W_CASES, ids=lambda c: "x".join(map(str, c)))
def test_conv3x3_winograd_f4_wgrad(case):
    """F(4x4,3x3) weight gradient through lsps_conv2d_wgrad against an f64 reference (own bound W4_TOL like the forward
    kernel; measured 2e-6 .. 5e-6), against the direct kernel, and the kernel the library reports."""
    _need_gpu()
    from lsps_amd import _lib, ops
    N, C, H, K = case
    L = _lib.lib()
    prev = ops.get_winograd()
    try:
        x = _rand(N, C, H, 32, seed=31)
        gy = _rand(N, K, H, 32, seed=32)
        w = torch.zeros(K, C, 3, 3, dtype=torch.float64, requires_grad=True)
        F.conv2d(x.double(), w, None, padding=1).backward(gy.double())
        xd, gyd = x.cuda(), gy.cuda()
        ws, wsb = _lib.workspace(L.lsps_conv2d_workspace_bytes(N, C, H, 32, K, 3, 3, 1, 1), xd.device)
        res = {}
        for mode in ('always', 'off'):
            ops.set_winograd(mode)
            dw = torch.full((K, C, 3, 3), float('nan'), device='cuda')
            _lib.check(L.lsps_conv2d_wgrad(_lib.ptr(xd), _lib.ptr(gyd), _lib.ptr(dw), None, N, C, H, 32, K, 3, 3, 1, 1, ws, wsb,
                                           _lib.stream()), 'wgrad')
            res[mode] = (dw, L.lsps_last_kernel(None))
        assert res['always'][1] == b'wino4_w3x3_kernel', res['always'][1]
        assert res['off'][1] != b'wino4_w3x3_kernel'
        assert _rel(res['always'][0], w.grad) < W4_TOL, _rel(res['always'][0], w.grad)
        assert _rel(res['always'][0], res['off'][0]) < W4_TOL
    finally:
        ops.set_winograd(prev)


@pytest.mark.parametrize("case", [(8, 256, 256, 4), (4, 256, 256, 8), (2, 256, 256, 8), (15, 256, 256, 2), (5, 128, 64, 4), (1, 64, 32, 2),
                                  (3, 192, 32, 6), (16, 256, 256, 0), (4, 32, 32, 0)], ids=lambda c: "x".join(map(str, c)))
@pytest.mark.parametrize("form", ["lrelu", "residual"])
def test_conv3x3_instance_norm_small_batch_reduction_split(case, form, monkeypatch):
    """The estimate modes run the generator on 4 + 4 samples (lsps_trainer.py:238): under the DEFAULT dispatch such launches take
    the F(4x4,3x3) kernel with the input channels split over `ks` workgroups per (image, k slice) and the partial outputs
    summed inside the InstanceNorm kernel (igemm.hip: wino4_split).  Against f64, incl. rstd; case = (N, C, K, expected ks;
    0 = the path must NOT be taken: grid already fills the chip / channels not splittable)."""
    _need_gpu()
    from lsps_amd import _lib, ops
    N, C, K, ks = case
    L = _lib.lib()
    assert ops.get_winograd() == 'auto'
    x = _rand(N, C, 32, 32, seed=31)
    w = _rand(K, C, 3, 3, seed=32, scale=0.1)
    res = _rand(N, K, 32, 32, seed=33) if form == 'residual' else None
    slope = 0.01 if form == 'lrelu' else -1.0
    c0 = F.conv2d(x.double(), w.double(), None, padding=1)
    ref = F.instance_norm(c0, eps=1e-5)
    ref = F.leaky_relu(ref, 0.01) if form == 'lrelu' else ref + res.double()
    rstd_ref = 1.0 / torch.sqrt(c0.var(dim=(2, 3), unbiased=False) + 1e-5)
    xd, wd = x.cuda(), w.cuda()
    rd = res.cuda() if res is not None else None
    ws, wsb = _lib.workspace(L.lsps_conv2d_workspace_bytes(N, C, 32, 32, K, 3, 3, 1, 1), xd.device)

    def run(entry=L.lsps_conv2d_in_fwd_nograd):
        y = torch.full((N, K, 32, 32), float('nan'), device='cuda')
        rstd = torch.empty(N * K, device='cuda')
        _lib.check(entry(_lib.ptr(xd), _lib.ptr(wd), _lib.ptr(rd), _lib.ptr(y), _lib.ptr(rstd), N, C, 32, 32, K,
                         slope, 1e-5, ws, wsb, _lib.stream()), 'conv2d_in_fwd')
        return y, rstd, [L.lsps_last_kernel(None).decode()]
    y, rstd, names = run()
    split = ks > 0
    assert ('wino4_f3x3_kernel' in names) == (split or N * (K // 32) >= 128), names
    assert _rel(y, ref) < W4_TOL, _rel(y, ref)
    assert _rel(rstd.view(N, K), rstd_ref) < 2e-5
    # the switch: LSPS_WINO4_SPLIT is read once per process, so compare against the composed path through the mode instead
    ops.set_winograd('off')
    try:
        y0, rstd0, names0 = run()
    finally:
        ops.set_winograd('auto')
    assert 'wino4_f3x3_kernel' not in names0
    assert _rel(y, y0.cpu()) < W4_TOL
    # the entry a differentiated pass calls never takes the split (the golden step cases at N = 2 pin its kernels)
    y1, rstd1, names1 = run(L.lsps_conv2d_in_fwd)
    assert ('wino4_f3x3_kernel' in names1) == (N * (K // 32) >= 128), names1
    assert _rel(y1, ref) < W4_TOL


@pytest.mark.parametrize("case", [(3, 256, 32, 256), (4, 64, 32, 64), (2, 8, 32, 32), (2, 64, 16, 64), (3, 24, 8, 40), (1, 16, 32, 48)],
                         ids=lambda c: "x".join(map(str, c)))
@pytest.mark.parametrize("form", ["lrelu", "residual", "plain"])
def test_conv3x3_instance_norm_fused(case, form):
    """lsps_conv2d_in_fwd = conv3x3 + InstanceNorm (+ LeakyReLU | + residual) (common_net.py:162-171, 177-181) against f64
    torch: the F(4x4,3x3) epilogue path on 32x32 maps with K % 32 == 0, the composed conv + norm pass elsewhere; rstd is
    what lsps_inorm_bwd expects."""
    _need_gpu()
    from lsps_amd import _lib, ops
    N, C, H, K = case
    L = _lib.lib()
    prev = ops.get_winograd()
    ops.set_winograd('always')
    try:
        x = _rand(N, C, H, 32, seed=21)
        w = _rand(K, C, 3, 3, seed=22, scale=0.1)
        res = _rand(N, K, H, 32, seed=23) if form == 'residual' else None
        slope = 0.01 if form == 'lrelu' else -1.0
        c0 = F.conv2d(x.double(), w.double(), None, padding=1)
        ref = F.instance_norm(c0, eps=1e-5)
        if form == 'lrelu':
            ref = F.leaky_relu(ref, 0.01)
        if res is not None:
            ref = ref + res.double()
        rstd_ref = 1.0 / torch.sqrt(c0.var(dim=(2, 3), unbiased=False) + 1e-5)
        xd, wd = x.cuda(), w.cuda()
        rd = res.cuda() if res is not None else None
        y = torch.empty(N, K, H, 32, device='cuda')
        rstd = torch.empty(N * K, device='cuda')
        ws, wsb = _lib.workspace(L.lsps_conv2d_workspace_bytes(N, C, H, 32, K, 3, 3, 1, 1), xd.device)
        _lib.check(L.lsps_conv2d_in_fwd(_lib.ptr(xd), _lib.ptr(wd), _lib.ptr(rd), _lib.ptr(y), _lib.ptr(rstd), N, C, H, 32, K,
                                        slope, 1e-5, ws, wsb, _lib.stream()), 'conv2d_in_fwd')
        f4 = H == 32 and K % 32 == 0 and C % 8 == 0
        assert (L.lsps_last_kernel(None) == b'wino4_f3x3_kernel') == f4
        assert _rel(y, ref) < W4_TOL, _rel(y, ref)
        assert _rel(rstd.view(N, K), rstd_ref) < 2e-5
        # the block's backward recovers x_hat from the OUTPUT and this rstd: run it against autograd of the f64 composition
        gy = _rand(N, K, H, 32, seed=24)
        c0g = c0.detach().clone().requires_grad_(True)
        r2 = F.instance_norm(c0g, eps=1e-5)
        if form == 'lrelu':
            r2 = F.leaky_relu(r2, 0.01)
        r2.backward(gy.double())
        dpre = torch.empty_like(y)
        _lib.check(L.lsps_inorm_bwd(_lib.ptr(gy.cuda()), _lib.ptr(y), _lib.ptr(rd), _lib.ptr(rstd), _lib.ptr(dpre), N * K, H * 32,
                                    slope, _lib.stream()), 'inorm_bwd')
        if form == 'lrelu':
            # where the normalised value is within round-off of 0 the LeakyReLU slope (1 vs 0.01) is decided by that
            # round-off: such isolated elements are excluded, everything else must agree
            xhat = F.instance_norm(c0, eps=1e-5)
            diff = (dpre.cpu().double() - c0g.grad).abs() / c0g.grad.abs().max()
            off = diff > LRELU_BWD_TOL
            assert int(off.sum()) <= 64 and bool((xhat[off].abs() < 1e-3).all()), (int(off.sum()), float(diff.max()))
        else:
            assert _rel(dpre, c0g.grad) < 1e-4, _rel(dpre, c0g.grad)
    finally:
        ops.set_winograd(prev)


# (N, C, H, W, K): 3x3 / stride 2 / pad 1 convs in batch-innermost layout (csrc/chwn.hip): the four discriminator-trunk
# geometries, small batches (reduction splits), a batch that is not a multiple of the 128-wide tile, non-square maps
CHWN_CASES = [
    (8, 128, 32, 32, 256),      # dis_s0 geometry: 16x16 outputs, wgrad over 256 positions split 29 ways
    (12, 256, 16, 16, 512),     # dis_s1
    (20, 512, 8, 8, 1024),      # dis_s2: forward / dgrad reduction splits
    (36, 1024, 4, 4, 2048),     # dis_s3: 2x2 outputs, 25 of 36 taps real
    (132, 128, 4, 6, 128),      # two n tiles (128 + 4), non-square map
    (8, 128, 12, 6, 128),       # 18 blocks of 2x2 input positions: not a multiple of 4 -> the dgrad keeps per-position workgroups
    (8, 128, 16, 8, 256),       # non-square map WITH grouped dgrad positions (32 blocks)
    (4, 128, 2, 2, 128),        # one output position, 4 of 9 taps real
]


@pytest.mark.parametrize("case", CHWN_CASES, ids=lambda c: "x".join(map(str, c)))
def test_conv3x3s2_chwn(case):
    """lsps_conv3x3s2_chwn_{fwd,dgrad,wgrad} + lsps_transpose2d against an f64 torch convolution (bias + LeakyReLU in the
    forward epilogue, LeakyReLU backward + bias gradient through lsps_act_bwd_bias on the [C][H][W][N] tensor)."""
    _need_gpu()
    from lsps_amd import ops
    N, C, H, W, K = case
    assert ops.conv3x3s2_chwn_ok(N, C, H, W, K)
    x = _rand(N, C, H, W, seed=51).double().requires_grad_(True)
    w = _rand(K, C, 3, 3, seed=52, scale=0.05).double().requires_grad_(True)
    b = _rand(K, seed=53, scale=0.1).double().requires_grad_(True)
    y_ref = F.leaky_relu(F.conv2d(x, w, b, stride=2, padding=1), 0.01)
    gy = _rand(*y_ref.shape, seed=54)
    y_ref.backward(gy.double())
    xd, wd, bd = (t.detach().float().cuda().requires_grad_(True) for t in (x, w, b))
    t = ops.nchw_to_chwn(xd)
    assert tuple(t.shape) == (C, H, W, N)
    assert torch.equal(t.detach().cpu(), xd.detach().cpu().permute(1, 2, 3, 0).contiguous())
    yc = ops.conv3x3s2_chwn(t, wd, bd, ops.ACT_LRELU, 0.01)
    y = ops.chwn_to_nchw(yc)
    assert _rel(y, y_ref) < 2e-5, _rel(y, y_ref)
    y.backward(gy.cuda())
    errs = dict(dx=_rel(xd.grad, x.grad), dw=_rel(wd.grad, w.grad), db=_rel(bd.grad, b.grad))
    assert errs['dx'] < LRELU_BWD_TOL and errs['dw'] < 1e-4 and errs['db'] < 1e-4, errs


def test_conv3x3s2_chwn_rejects_unsupported_geometry():
    _need_gpu()
    from lsps_amd import _lib, ops
    L = _lib.lib()
    assert not ops.conv3x3s2_chwn_ok(6, 128, 8, 8, 256)        # N % 4 != 0
    assert not ops.conv3x3s2_chwn_ok(8, 64, 8, 8, 256)         # C % 128 != 0
    assert not ops.conv3x3s2_chwn_ok(8, 128, 7, 8, 256)        # odd map
    x = torch.zeros(128, 8, 8, 6, device='cuda')
    w = torch.zeros(256, 128, 3, 3, device='cuda')
    y = torch.zeros(256, 4, 4, 6, device='cuda')
    ws, wsb = _lib.workspace(1 << 20, x.device)
    assert L.lsps_conv3x3s2_chwn_fwd(_lib.ptr(x), _lib.ptr(w), None, _lib.ptr(y), 6, 128, 8, 8, 256, 0, 0.0, ws, wsb,
                                     _lib.stream()) != 0
    assert b'geometry' in L.lsps_last_error()


_VARIANT_SNIPPET = r"""
import sys, torch, torch.nn.functional as F
sys.path.insert(0, %(repo)r)
from lsps_amd import _lib, ops
L = _lib.lib()
torch.manual_seed(0)
def rel(a, b):
    return float((a.double().cpu() - b.double().cpu()).abs().max() / b.double().abs().max())
# 3x3 / stride-2 forward (igemm_f3x3s2_kernel, LSPS_FS2_CC)
x = torch.randn(3, 64, 64, 64, device='cuda'); w = torch.randn(128, 64, 3, 3, device='cuda') * 0.05; b = torch.randn(128, device='cuda')
y = ops.conv2d(x, w, b, 2, 1, ops.ACT_LRELU, 0.01)
assert L.lsps_last_kernel(None) == b'igemm_f3x3s2_kernel', L.lsps_last_kernel(None)
ref = F.leaky_relu(F.conv2d(x.double().cpu(), w.double().cpu(), b.double().cpu(), stride=2, padding=1), 0.01)
assert rel(y, ref) < 2e-5, rel(y, ref)
# F(4x4,3x3) weight gradient (wino4_w3x3_kernel, LSPS_WINO4W_WAVES)
ops.set_winograd('always')
N, C, K = 5, 64, 128
x = torch.randn(N, C, 32, 32, device='cuda'); gy = torch.randn(N, K, 32, 32, device='cuda')
dw = torch.empty(K, C, 3, 3, device='cuda')
ws, wsb = _lib.workspace(L.lsps_conv2d_workspace_bytes(N, C, 32, 32, K, 3, 3, 1, 1), x.device)
_lib.check(L.lsps_conv2d_wgrad(_lib.ptr(x), _lib.ptr(gy), _lib.ptr(dw), None, N, C, 32, 32, K, 3, 3, 1, 1, ws, wsb, _lib.stream()), 'wgrad')
assert L.lsps_last_kernel(None) == b'wino4_w3x3_kernel'
wr = torch.zeros(K, C, 3, 3, dtype=torch.float64, requires_grad=True)
F.conv2d(x.double().cpu(), wr, None, padding=1).backward(gy.double().cpu())
assert rel(dw, wr.grad) < 5e-5, rel(dw, wr.grad)
print('variants ok')
"""


@pytest.mark.parametrize("env", [{'LSPS_FS2_CC': '8', 'LSPS_WINO4W_WAVES': '8'}, {'LSPS_FS2_CC': '4', 'LSPS_WINO4W_WAVES': '4'}],
                         ids=['cc8_waves8', 'cc4_waves4'])
def test_env_selected_kernel_variants(env):
    """The A/B knobs that are read once per process (LSPS_FS2_CC: 4- / 8-channel chunks of the stride-2 forward kernel;
    LSPS_WINO4W_WAVES: 4- / 8-wave workgroups of the F(4x4,3x3) weight gradient): both settings of each give the same
    results against f64 references, in a fresh process."""
    _need_gpu()
    import os
    import subprocess
    import sys
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    e = dict(os.environ)
    e.update(env)
    out = subprocess.run([sys.executable, '-c', _VARIANT_SNIPPET % dict(repo=repo)], env=e, capture_output=True, text=True,
                         timeout=600)
    assert out.returncode == 0 and 'variants ok' in out.stdout, (out.stdout[-400:], out.stderr[-1200:])



@pytest.mark.parametrize("case", [(2, 1, 128, 128, 64, 7, 1, 3), (3, 1, 128, 128, 64, 7, 2, 3), (3, 1, 32, 64, 40, 5, 1, 2)],
                         ids=lambda c: "x".join(map(str, c)))
def test_stem_weight_gradient_with_fused_activation_backward(case):
    """One-input-channel stems whose image needs no gradient: dw and db from ONE kernel (LeakyReLU backward applied while dy is
    staged, bias gradient as an extra tap column; csrc/conv_c1.h) against torch on the SAME activation mask (taken from the
    kernel's own forward output, so that no pre-activation that is zero to round-off decides the comparison)."""
    _need_gpu()
    from lsps_amd import ops
    N, C, H, W, K, R, st, pad = case
    x = _rand(N, C, H, W, seed=1)
    w = _rand(K, C, R, R, seed=2, scale=0.1)
    b = _rand(K, seed=3, scale=0.1)
    xd = x.cuda()
    wd, bd = (t.cuda().requires_grad_(True) for t in (w, b))
    ops.kernel_log_begin()
    y = ops.conv2d(xd, wd, bd, st, pad, ops.ACT_LRELU, 0.01)
    gy = _rand(*y.shape, seed=4)
    y.backward(gy.cuda())
    names = ops.kernel_log_end()
    assert names.count('c1_wgrad_kernel') == 1 and 'igemm_w_kernel' not in names, names
    g = torch.where(y.detach().cpu() > 0, gy, gy * 0.01).double()
    dw_ref = torch.nn.grad.conv2d_weight(x.double(), (K, C, R, R), g, stride=st, padding=pad)
    assert _rel(wd.grad, dw_ref) < 2e-4 and _rel(bd.grad, g.sum((0, 2, 3))) < 2e-4


def test_output_head_dgrad_with_fused_activation_backward(monkeypatch):
    """Decoder tail in f32 (LeakyReLUConvTranspose2d -> ConvTranspose2d(64, 1, 1) + Tanh through run_layers): the head's input
    gradient carries the previous layer's LeakyReLU backward and bias gradient; against the separate-pass path."""
    _need_gpu()
    from lsps_amd import ops
    from lsps_amd.trainers import common_net as cn
    torch.manual_seed(5)
    dev = torch.device('cuda')
    layers = [cn.LeakyReLUConvTranspose2d(128, 64, 3, 2, 1, 1), cn.ConvTranspose2d(64, 1, 1, 1, 0, act=cn.ACT_TANH)]
    for m in layers:
        m.to(dev)
    z = torch.randn(3, 128, 64, 64, device=dev)
    res = []
    for fuse in ('1', '0'):
        monkeypatch.setattr(ops.options, '_current', ops.options.from_env({'LSPS_FUSE_ACT': fuse}))
        for m in layers:
            for p in m.parameters():
                p.grad = None
        zz = z.clone().requires_grad_(True)
        out = cn.run_layers(layers, zz)
        (out.square().mean() + out.abs().mean()).backward()
        res.append([zz.grad.clone()] + [p.grad.clone() for m in layers for p in m.parameters()])
    for a, b in zip(*res):
        assert float((a - b).abs().max()) <= 1e-5 * float(b.abs().max()) + 1e-12, (a.shape, float((a - b).abs().max()), float(b.abs().max()))
