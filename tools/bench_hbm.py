#!/usr/bin/env python
"""HBM roofline of the HBM-bound kernels of the depth path (stem 7x7, 1x1 output + tanh, the
InstanceNorm / LeakyReLU / residual passes, loss reductions, Adam), at the shapes of the bs=128 pretrain step.
achieved = compulsory bytes (each operand read once, each result written once) / HIP-event time; peak 8 TB/s.
usage: python tools/bench_hbm.py [--iters 20] > profiles/<round>_hbm_kernels.json"""
import argparse
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from lsps_amd import ops  # noqa: E402
from lsps_amd.optim import FlatAdam  # noqa: E402

PEAK = 8000.0


def time_it(fn, iters):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--iters', type=int, default=20)
    a = ap.parse_args()
    dev = torch.device('cuda', 0)
    R = []

    def rec(name, what, nbytes, ms):
        r = {'kernel': name, 'op': what, 'compulsory_MB': nbytes / 1e6, 'ms': ms, 'GB/s': nbytes / ms / 1e6,
             'frac_of_hbm_peak': nbytes / ms / 1e6 / PEAK}
        R.append(r)
        print(json.dumps(r))

    g = lambda *s: torch.randn(*s, device=dev)          # noqa: E731
    N = 128
    # ---- InstanceNorm (+LeakyReLU / +residual), in place: [2N, 256, 32, 32]
    t = g(2 * N, 256, 32, 32)
    nb = t.numel() * 4
    with torch.no_grad():
        rec('inorm_fwd_kernel', 'IN + LeakyReLU in place, [256,256,32,32]', 2 * nb, time_it(lambda: ops.instance_norm_(t, None, 0.01), a.iters))
        res = g(2 * N, 256, 32, 32)
        rec('inorm_fwd_kernel', 'IN + residual add in place', 3 * nb, time_it(lambda: ops.instance_norm_(t, res, -1.0), a.iters))
    y = g(2 * N, 256, 32, 32).requires_grad_(True)
    out = ops.instance_norm_(y.clone(), None, 0.01)
    go = g(2 * N, 256, 32, 32)
    rec('inorm_bwd_kernel', 'IN + LeakyReLU backward from the output (reads dout, out; writes dy)', 3 * nb,
        time_it(lambda: torch.autograd.grad(out, y, go, retain_graph=True), a.iters))
    # ---- activation backward + bias gradient of the stem output [N, 64, 128, 128]
    from lsps_amd import _lib
    L = _lib.lib()
    dy, yy, dx = g(N, 64, 128, 128), g(N, 64, 128, 128), torch.empty(N, 64, 128, 128, device=dev)
    db = torch.empty(64, device=dev)
    ws, wsb = _lib.workspace(1 << 20, dev)
    nb = dy.numel() * 4
    rec('act_bwd_bias_kernel', 'LeakyReLU backward + bias gradient, [128,64,128,128]', 3 * nb,
        time_it(lambda: _lib.check(L.lsps_act_bwd_bias(dy.data_ptr(), yy.data_ptr(), dx.data_ptr(), db.data_ptr(), N, 64,
                                                        128 * 128, 1, 0.01, ws, wsb, _lib.stream()), 'x'), a.iters))
    # ---- stem 7x7 forward / wgrad, 1x1 output head
    x1, w7, b7 = g(N, 1, 128, 128), g(64, 1, 7, 7) * 0.05, g(64)
    with torch.no_grad():
        rec('c1_fwd_kernel', '7x7 stem forward 1->64 @128x128 (writes 64x128x128 per sample)', N * (1 + 64) * 16384 * 4,
            time_it(lambda: ops.conv2d(x1, w7, b7, 1, 3, ops.ACT_LRELU, 0.01), a.iters))
    xs = x1.clone().requires_grad_(False)
    w7g = w7.clone().requires_grad_(True)
    ys = ops.conv2d(xs, w7g, None, 1, 3)
    gy = g(N, 64, 128, 128)
    rec('c1_wgrad_kernel', '7x7 stem weight gradient (reads dy 64x128x128 per sample)', N * (1 + 64) * 16384 * 4,
        time_it(lambda: torch.autograd.grad(ys, w7g, gy, retain_graph=True), a.iters))
    xo, wo, bo = g(2 * N, 64, 128, 128), g(64, 1, 1, 1) * 0.05, g(1)
    with torch.no_grad():
        rec('pw1_fwd_kernel', '1x1 ConvTranspose 64->1 + tanh (reads 64x128x128 per sample)', 2 * N * (64 + 1) * 16384 * 4,
            time_it(lambda: ops.conv_transpose2d(xo, wo, bo, 1, 0, 0, ops.ACT_TANH), a.iters))
    # ---- losses
    p1, p2 = g(N, 1, 128, 128), g(N, 1, 128, 128)
    with torch.no_grad():
        rec('loss_partial_kernel', 'L1 loss of two [128,1,128,128] images', 2 * p1.numel() * 4, time_it(lambda: ops.l1_loss(p1, p2), a.iters))
        sh = g(2 * N, 256, 32, 32)
        rec('loss_partial_kernel', 'KL term mean(mu^2) over the shared latent [256,256,32,32]', sh.numel() * 4,
            time_it(lambda: ops.kl_loss(sh), a.iters))
    # ---- Adam over a discriminator-sized arena (25.4 M parameters): p, g, m, v read; p, m, v written
    ps = [torch.nn.Parameter(g(25_400_000 // 4, 4))]
    opt = FlatAdam(ps, lr=1e-4, betas=(0.5, 0.999), weight_decay=1e-4)
    opt.attach()
    ps[0].grad.normal_()

    def adam():
        opt.arena.touched = [True]
        opt.step()
    rec('adam_kernel', 'Adam step over 25.4 M parameters (7 x 4 B per parameter)', 7 * 4 * ps[0].numel(), time_it(adam, a.iters))
    # ---- noise add
    nz = g(2 * N, 256, 32, 32)
    with torch.no_grad():
        rec('axpy_kernel', 'GaussianNoiseLayer x + noise on [256,256,32,32]', 3 * nz.numel() * 4, time_it(lambda: ops.axpy(t, nz, 1.0), a.iters))
    return R


if __name__ == '__main__':
    main()
