"""The block classes of common_net.py that no shipped config instantiates (BatchNorm variants, ReLU-after-InstanceNorm
variants, the convolutional VAE head, LeakyReLUResBlock) on the HIP kernels, against vectors produced by the REFERENCE's
own classes (tests/golden/golden_blocks.npz, made by `tests/golden/make_golden.py blocks`): training-mode forward and
backward twice (running statistics move), parameter gradients, the buffers, eval-mode forward."""
import os

import numpy as np
import pytest
import torch

import cases

pytestmark = pytest.mark.gpu
G = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'golden_blocks.npz'))


@pytest.mark.parametrize('name', list(cases.BLOCK_CASES))
def test_block_matches_reference(name):
    if not torch.cuda.is_available():
        pytest.skip("no HIP device")
    import lsps_amd.trainers as prod
    got = cases.run_block_case(prod, name, lambda t: t.cuda(), lambda t: t.detach().cpu().numpy().copy())
    want_keys = [k for k in G.files if k.startswith(name + '/')]
    wscale = max(float(np.abs(G[k]).max()) for k in want_keys if k.endswith('.weight') and '/grad/' in k)
    # A conv / linear bias that feeds a batch- or instance-norm is cancelled by the mean subtraction: its gradient is
    # exactly 0, the reference computes round-off there; the InstanceNorm blocks here do not even apply it.
    dead = [k for k in want_keys if '/grad/' in k and k.endswith('.bias') and float(np.abs(G[k]).max()) <= 1e-5 * wscale]
    for k in dead:
        assert k not in got or float(np.abs(got[k]).max()) <= 1e-4 * wscale, k
    want_keys = [k for k in want_keys if k not in dead]
    assert sorted(k for k in got if k not in dead) == sorted(want_keys)
    for k in want_keys:
        w, g = G[k], got[k]
        scale = max(float(np.abs(w).max()), 1e-6)
        err = float(np.abs(g - w).max()) / scale
        # forward / buffers tight; gradients through ReLU / LeakyReLU kinks looser (a pre-activation within rounding
        # distance of 0 flips one element's slope)
        tol = 2e-4 if ('/y' in k or '/buffers/' in k) else 5e-3
        assert err <= tol, (k, err)


def test_gaussian_smoother():
    """GaussianSmoother (common_net.py:12-30): cv2.getGaussianKernel's published rule + replicate padding; against
    torch's conv2d with the same kernel."""
    if not torch.cuda.is_available():
        pytest.skip("no HIP device")
    import torch.nn.functional as F
    import lsps_amd.trainers as prod
    from lsps_amd.trainers.common_net import gaussian_kernel_1d
    assert np.allclose(gaussian_kernel_1d(5), [0.0625, 0.25, 0.375, 0.25, 0.0625])
    k9 = gaussian_kernel_1d(9)
    assert abs(k9.sum() - 1) < 1e-12 and np.allclose(k9, k9[::-1]) and abs(k9[4] / k9[3] - np.exp(0.5 / 1.7 ** 2)) < 1e-12
    for ks in (5, 9):
        sm = prod.GaussianSmoother(ks)
        sm.cuda(0)
        x = torch.as_tensor(cases.noise((3, 1, 40, 36), 5))
        want = F.conv2d(F.pad(x, [sm.pad] * 4, mode='replicate'), sm.blur_kernel.cpu())
        got = sm(x.cuda()).cpu()
        assert got.shape == want.shape == x.shape
        assert float((got - want).abs().max()) < 1e-5
