"""The CPU oracle (oracle/lsps_ref.py) against golden vectors captured from the REAL reference
(tests/golden/make_golden.py).  This is what pins the oracle: every public method of the nets
and every update step, tiny (ch=8/4) and full (ch=64) width.  Tolerance: 1e-4 relative to the
tensor's abs-max (same torch CPU kernels, different op composition => ~1e-6 expected)."""
import numpy as np
import pytest
import torch

import cases
from oracle import lsps_ref

RTOL = 1e-4


@pytest.mark.parametrize("config", ["tiny", "full"])
def test_oracle_modules_match_reference(config, golden):
    torch.set_num_threads(8)
    A = cases.NativeAdapter(lsps_ref, 'cpu')
    R = cases.run_module_cases(A, config, lsps_ref)
    g = {k: v for k, v in golden(config).items() if k.split('/')[0] in R}
    assert g, "no golden entries"
    bad, worst = cases.compare(R, g, RTOL)
    assert not bad, "worst=%g first failures: %s" % (worst, bad[:5])


@pytest.mark.parametrize("config", ["tiny", "full"])
def test_oracle_steps_match_reference(config, golden):
    torch.set_num_threads(8)
    A = cases.NativeAdapter(lsps_ref, 'cpu')
    R = cases.run_step_cases(A, config, lsps_ref)
    g = {k: v for k, v in golden(config).items() if k.split('/')[0] in R}
    assert g, "no golden entries"
    # Adam divides by sqrt(v)+eps: where a gradient is ~0 the update direction is noise-dominated,
    # so post-step weights are compared at 1e-3 of the tensor's abs-max (weights ~0.02, lr 1e-4).
    bad, worst = cases.compare(R, g, 1e-3, grad_rtol=2e-2)
    assert not bad, "worst=%g first failures: %s" % (worst, bad[:5])


def test_oracle_extra_cases_match_reference(golden):
    """post_update(mode=1) (tiny + full) and the full-width pretrain iteration at 3 samples per domain."""
    torch.set_num_threads(8)
    A = cases.NativeAdapter(lsps_ref, 'cpu')
    R = cases.run_extra_cases(A, lsps_ref)
    g = golden('extra')
    assert set(k.split('/')[0] for k in g) == set(R)
    bad, worst = cases.compare(R, g, 1e-3, grad_rtol=2e-2)
    assert not bad, "worst=%g first failures: %s" % (worst, bad[:5])


def test_oracle_resnext_generator_matches_reference(golden):
    torch.set_num_threads(8)
    A = cases.NativeAdapter(lsps_ref, 'cpu')
    R = cases.run_resx_cases(A, lsps_ref)
    g = {k: v for k, v in golden('tiny').items() if k.split('/')[0] in R}
    assert g, "no golden entries"
    # three LeakyReLU/InstanceNorm stages per block at 8-64 channels: gradients are even more kink-sensitive
    bad, worst = cases.compare(R, g, 1e-3, grad_rtol=5e-2, grad_robust=cases.GRAD_ROBUST_RESX)
    assert not bad, "worst=%g first failures: %s" % (worst, bad[:5])


def test_oracle_expand_layer_discriminator_matches_reference(golden):
    """`SharedDis` with `n_expand_layer: 1` (lsps_nets.py:93,116-118; optional key, unused by both YAMLs): the oracle's
    stride-1 expand conv against vectors captured from the reference's own class (golden_expand.npz)."""
    torch.set_num_threads(8)
    A = cases.NativeAdapter(lsps_ref, 'cpu')
    R = cases.run_expand_cases(A, lsps_ref)
    g = golden('expand')
    assert set(k.split('/')[0] for k in g) == set(R)
    assert tuple(g['expand.dis.forward/feats_a/shape']) == (2, 256, 2, 2)         # 4 -> 8 -> (expand) 16 -> ... -> 256
    assert tuple(g['expand.dis.regress_a.n1/p0/shape']) == (20,)                  # .squeeze() at n = 1 (lsps_nets.py:139)
    bad, worst = cases.compare(R, g, 1e-3, grad_rtol=2e-2)
    assert not bad, "worst=%g first failures: %s" % (worst, bad[:5])


def test_shape_tables_match_reference_keys(golden):
    """State-dict key sets / shapes equal the reference's (80/20/10/8 tensors)."""
    hp = cases.hp_for('full')
    assert len(lsps_ref.gen_shapes(hp['gen'])) == 80
    assert len(lsps_ref.dis_shapes(hp['dis'])) == 20
    assert len(lsps_ref.vae_shapes(hp['vae'])) == 10
    assert len(lsps_ref.map_shapes(hp['map'])) == 8
    g = golden('full')
    for net, shapes in (('gen', lsps_ref.gen_shapes(hp['gen'])), ('dis', lsps_ref.dis_shapes(hp['dis']))):
        for k, s in shapes.items():
            assert tuple(g['pretrain.it0.%s.params/%s/shape' % (net, k)]) == tuple(s)


def test_resblock_dropout_matches_reference():
    """`res_dropout_ratio` > 0 (lsps_nets.py:176-179): the oracle's residual block with the recorded keep mask against
    the reference's own LeakyINSResBlock(dropout=p) (golden_dropout.npz), forward and backward, train and eval."""
    import os
    G = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'golden_dropout.npz'))
    d = cases.dropout_case_inputs()
    p = {'b.model.0.weight': torch.as_tensor(d['w0']).requires_grad_(True), 'b.model.0.bias': torch.as_tensor(d['b0']),
         'b.model.3.weight': torch.as_tensor(d['w3']).requires_grad_(True), 'b.model.3.bias': torch.as_tensor(d['b3'])}
    x = torch.as_tensor(d['x']).clone().requires_grad_(True)
    y = lsps_ref.leaky_ins_res_block(x, p, 'b', drop_mask=torch.as_tensor(d['mask']))
    y.backward(torch.as_tensor(d['gy']))
    for name, got in (('y', y), ('dx', x.grad), ('dw0', p['b.model.0.weight'].grad), ('dw3', p['b.model.3.weight'].grad)):
        want = G['drop.train.' + name]
        assert np.abs(got.detach().numpy() - want).max() <= 1e-5 * max(1.0, np.abs(want).max()), name
    with torch.no_grad():
        ye = lsps_ref.leaky_ins_res_block(torch.as_tensor(d['x']), p, 'b')
    assert np.abs(ye.numpy() - G['drop.eval.y']).max() <= 1e-5
