"""End-to-end parity of the HIP product path (lsps_amd.trainers, through the C-ABI) against
  (a) the golden vectors captured from the REAL reference (tests/golden/golden_*.npz), and
  (b) the CPU oracle run on the same seeded inputs here.
Tolerances (north_star: 1e-3 rel fp32): forward tensors and loss scalars 1e-3 of abs-max;
gradients 2e-2 (discontinuous in the activations, see cases.compare); post-Adam weights via the
robust rule in cases.compare."""
import numpy as np
import pytest
import torch

import cases
from oracle import lsps_ref

pytestmark = pytest.mark.gpu
RTOL = 1e-3


def _adapter():
    if not torch.cuda.is_available():
        pytest.skip("no HIP device")
    import lsps_amd.trainers as prod
    return cases.NativeAdapter(prod, 'cuda')


@pytest.mark.parametrize("config", ["tiny", "full"])
def test_modules_match_reference_golden(config, golden):
    A = _adapter()
    R = cases.run_module_cases(A, config, lsps_ref)
    g = {k: v for k, v in golden(config).items() if k.split('/')[0] in R}
    bad, worst = cases.compare(R, g, RTOL)
    print("worst rel err", worst)
    assert not bad, "worst=%g first failures: %s" % (worst, bad[:8])


@pytest.mark.parametrize("config", ["tiny", "full"])
def test_steps_match_reference_golden(config, golden):
    A = _adapter()
    R = cases.run_step_cases(A, config, lsps_ref)
    g = {k: v for k, v in golden(config).items() if k.split('/')[0] in R}
    bad, worst = cases.compare(R, g, RTOL, grad_rtol=2e-2)
    print("worst rel err", worst)
    assert not bad, "worst=%g first failures: %s" % (worst, bad[:8])


@pytest.mark.parametrize("mode", ["always", "off", "x3"])
def test_full_steps_match_reference_golden_per_conv_algorithm(mode, golden, monkeypatch):
    """The `full` config's update steps against the REAL reference's golden vectors with the residual convs forced
    onto the Winograd kernels (forward, dgrad and weight gradient, also at this small batch) and onto the direct ones:
    both sit inside the same tolerances ('auto', the default, is what the test above runs).  In the 'always' run the
    discriminator trunk is also forced onto the batch-innermost kernels the bench-size batches use (csrc/chwn.hip; their
    batch threshold is lifted), so the whole default bs=128 dispatch is checked against the reference at trainer level.
    'x3' (round 5): every 3x3 / stride-2 conv and transposed conv of both nets on the three-limb kernels of csrc/x3s2.h (their
    work threshold is lifted: at bench sizes they are the default dispatch of those layers), everything else as 'auto'."""
    A = _adapter()
    from lsps_amd import ops
    prev = ops.get_winograd()
    ops.set_winograd(mode if mode != 'x3' else 'auto')
    env = {'LSPS_X3_MIN_GMAC': '0'} if mode == 'x3' else {'LSPS_CHWN_MIN_N': '1' if mode == 'always' else '1000000', 'LSPS_X3': '0'}
    monkeypatch.setattr(ops.options, '_current', ops.options.from_env(env))
    ops.kernel_log_begin()
    try:
        R = cases.run_step_cases(A, 'full', lsps_ref)
    finally:
        names = ops.kernel_log_end()
        ops.set_winograd(prev)
    assert ('chwn_gemm_kernel' in names and 'chwn_wgrad_kernel' in names) == (mode == 'always'), sorted(set(names))
    assert all((k in names) == (mode == 'x3') for k in ('x3s2_fwd_kernel', 'x3s2_tr_kernel', 'x3s2_wgrad_kernel')), sorted(set(names))
    if mode == 'x3':
        assert not [k for k in names if k.startswith('igemm_') and '3x3s2' in k], sorted(set(names))
    if mode == 'always':
        assert 'wino4_f3x3_kernel' in names and 'wino4_w3x3_kernel' in names, sorted(set(names))
    g = {k: v for k, v in golden('full').items() if k.split('/')[0] in R}
    bad, worst = cases.compare(R, g, RTOL, grad_rtol=2e-2)
    print("worst rel err", worst)
    assert not bad, "worst=%g first failures: %s" % (worst, bad[:8])


def test_n16_full_width_default_dispatch_against_the_oracle(monkeypatch):
    """a DIRECT oracle-vs-HIP comparison at a batch where the default dispatch is the bench's: full width,
    16 samples per domain, `dis_update` + `gen_update` + `post_update(3)` once.  Forward tensors and loss scalars at 1e-3 of
    abs-max, gradients through the robust criteria of cases.compare; the kernel log must show the F(4x4,3x3) forward / dgrad and
    weight-gradient kernels, the stride-2 family (three-limb kernels where the work threshold routes to them, the exact-f32
    ones elsewhere) and the batch-innermost trunk (its batch threshold is lowered to this batch: the 16- and 32-image passes
    of post_update / gen_update; dis_update's 96-image pass is above the three-limb family's work threshold, set to 4 x 10^9
    multiply-adds here so that all of these families appear at this batch)."""
    A = _adapter()
    from lsps_amd import ops
    import os
    torch.set_num_threads(max(1, (os.cpu_count() or 2) // 2))
    O = cases.NativeAdapter(lsps_ref, 'cpu')
    gold = cases.flatten(cases.run_n16_cases(O, lsps_ref))
    monkeypatch.setattr(ops.options, '_current', ops.options.from_env({'LSPS_CHWN_MIN_N': '16', 'LSPS_X3_MIN_GMAC': '4'}))
    ops.kernel_log_begin()
    try:
        R = cases.run_n16_cases(A, lsps_ref)
    finally:
        names = set(ops.kernel_log_end())
    for k in ('wino4_f3x3_kernel', 'wino4_w3x3_kernel', 'chwn_gemm_kernel', 'chwn_wgrad_kernel', 'x3s2_fwd_kernel', 'x3s2_tr_kernel',
              'x3s2_wgrad_kernel'):
        assert k in names, (k, sorted(names))
    assert [k for k in names if k.startswith('igemm_') and '3x3s2' in k], sorted(names)
    report = {}
    bad, worst = cases.compare(R, gold, RTOL, grad_rtol=2e-2, report=report)
    print("worst rel err", worst, report)
    assert not bad, "worst=%g first failures: %s" % (worst, bad[:8])


def test_n32_full_width_literal_default_dispatch_against_the_oracle():
    """VERDICT r5 "what's weak" 2: the N = 16 test above moves two thresholds so that every kernel family appears, i.e. it is A
    mixed dispatch, not the bench's.  Here NOTHING is overridden: full width, 32 samples per domain, `dis_update` + `gen_update` +
    `post_update(3)` once against the CPU oracle run in the test (~1.5 min on the GPU box's host).  At this batch the default
    dispatch is the bs = 128 bench's: F(4x4,3x3) forward / dgrad / wgrad for the residual convs, the three-limb kernels for EVERY
    stride-2 layer of both nets (>= 10^9 multiply-adds per launch), no batch-innermost trunk."""
    A = _adapter()
    from lsps_amd import ops
    import os
    torch.set_num_threads(max(1, (os.cpu_count() or 2) // 2))
    assert ops.options.get() == ops.options.from_env({}) and ops.get_winograd() == 'auto'      # the literal defaults
    O = cases.NativeAdapter(lsps_ref, 'cpu')
    gold = cases.flatten(cases.run_n16_cases(O, lsps_ref, n=32))
    ops.kernel_log_begin()
    try:
        R = cases.run_n16_cases(A, lsps_ref, n=32)
    finally:
        names = set(ops.kernel_log_end())
    for k in ('wino4_f3x3_kernel', 'wino4_w3x3_kernel', 'x3s2_fwd_kernel', 'x3s2_tr_kernel', 'x3s2_wgrad_kernel'):
        assert k in names, (k, sorted(names))
    # no batch-innermost trunk at this batch.  (Exact-f32 stride-2 kernels DO appear, as in the bs = 128 estimate3 step: the merged
    # discriminator pass of post_update runs front B on the 4 + 4 generated samples only, 0.6 x 10^9 multiply-adds, below the family's
    # work threshold; the pretrain iteration has no such launch.)
    assert not [k for k in names if k.startswith('chwn_')], sorted(names)
    report = {}
    bad, worst = cases.compare(R, gold, RTOL, grad_rtol=2e-2, report=report)
    print("worst rel err", worst, report)
    assert not bad, "worst=%g first failures: %s" % (worst, bad[:8])


def test_extra_cases_match_reference_golden(golden):
    """`post_update(mode=1)` (lsps_trainer.py:231-234) and a full-width pretrain iteration at a batch where the DEFAULT
    dispatch ('auto') of the residual convs is the Winograd path — the dispatch the bench runs, at trainer level."""
    A = _adapter()
    from lsps_amd import ops
    assert ops.get_winograd() == 'auto'
    ops.kernel_log_begin()
    try:
        R = cases.run_extra_cases(A, lsps_ref)
    finally:
        names = ops.kernel_log_end()
    assert any(n.startswith('wino') for n in names), sorted(set(names))
    bad, worst = cases.compare(R, golden('extra'), RTOL, grad_rtol=2e-2)
    print("worst rel err", worst)
    assert not bad, "worst=%g first failures: %s" % (worst, bad[:8])


def test_expand_layer_discriminator_matches_reference_golden(golden):
    """`n_expand_layer: 1` (lsps_nets.py:93,116-118): the product's stride-1 expand conv in front of the trunk, module
    outputs + dis_update / post_update(3) x 2 against the reference's own vectors (golden_expand.npz)."""
    A = _adapter()
    R = cases.run_expand_cases(A, lsps_ref)
    g = golden('expand')
    assert set(k.split('/')[0] for k in g) == set(R)
    bad, worst = cases.compare(R, g, RTOL, grad_rtol=2e-2)
    print("worst rel err", worst)
    assert not bad, "worst=%g first failures: %s" % (worst, bad[:8])


def test_resnext_generator_matches_reference_golden(golden):
    A = _adapter()
    R = cases.run_resx_cases(A, lsps_ref)
    g = {k: v for k, v in golden('tiny').items() if k.split('/')[0] in R}
    bad, worst = cases.compare(R, g, RTOL, grad_rtol=5e-2, grad_robust=cases.GRAD_ROBUST_RESX)
    assert g and not bad, "worst=%g first failures: %s" % (worst, bad[:8])


def test_joint_readout_matches_oracle():
    """A12: regress_b -> vae.decode -> mm joints; worst-joint argmax and <=40 mm decisions identical,
    joint coordinates within 1e-3 (depth_train.py:200-253, handpose_evaluation.py:97,130-136,203)."""
    A = _adapter()
    hp = cases.hp_for('full')
    sds = cases.make_weights(hp, lsps_ref)
    n = 16
    b = cases.make_inputs(n)
    cube = np.array([300.0, 300.0, 300.0], np.float32)
    ref_tr = cases.NativeAdapter(lsps_ref, 'cpu').make_trainer(hp, sds)
    ref = lsps_ref.joint_readout(ref_tr.dis, ref_tr.vae, torch.as_tensor(b['xb']), torch.as_tensor(b['lb']), b['cb'], cube)
    tr = A.make_trainer(hp, sds)
    tr.dis.eval()
    with torch.no_grad():
        _, post, _ = tr.dis.regress_b(A.T(b['xb']))
        pose = tr.vae.decode(post).cpu().numpy()
    gt = b['lb'].reshape(n, -1, 3)[:, lsps_ref.NYU_EVAL_JOINTS]
    pr = pose.reshape(n, -1, 3)[:, lsps_ref.NYU_EVAL_JOINTS]
    com = b['cb'].reshape(n, 1, 3)
    pr3d, gt3d = pr * (cube[0] / 2.) + com, gt * (cube[0] / 2.) + com
    err = np.sqrt(np.square(gt3d - pr3d).sum(axis=2))
    assert np.abs(pose - ref['pose']).max() <= 1e-3 * np.abs(ref['pose']).max()
    assert (np.argmax(err, axis=1) == ref['worst_joint']).all()
    assert int((np.nanmax(err, axis=1) <= 40).sum()) == ref['frames_within_40']
    assert abs(np.nanmean(np.nanmean(err, axis=1)) - ref['mean_err']) <= 1e-3 * ref['mean_err']


def test_joint_readout_bf16_math_mode():
    """A12 in the bf16 math mode (BASELINE config 5; the tolerance that matters there): regress_b on the C8 bf16
    stride-2 kernels -> vae.decode -> mm joints against the f32 CPU oracle: joint coordinates within 1e-2 of their abs-max,
    the worst joint per frame and the <= 40 mm decisions identical."""
    from lsps_amd import ops
    A = _adapter()
    hp = cases.hp_for('full')
    sds = cases.make_weights(hp, lsps_ref)
    n = 16
    b = cases.make_inputs(n)
    cube = np.array([300.0, 300.0, 300.0], np.float32)
    ref_tr = cases.NativeAdapter(lsps_ref, 'cpu').make_trainer(hp, sds)
    ref = lsps_ref.joint_readout(ref_tr.dis, ref_tr.vae, torch.as_tensor(b['xb']), torch.as_tensor(b['lb']), b['cb'], cube)
    tr = A.make_trainer(hp, sds)
    tr.dis.eval()
    ops.set_math_mode('bf16')
    try:
        ops.kernel_log_begin()
        with torch.no_grad():
            _, post, _ = tr.dis.regress_b(A.T(b['xb']))
            pose = tr.vae.decode(post).cpu().numpy()
        ops.kernel_log_end()
    finally:
        ops.set_math_mode('f32')
    gt = b['lb'].reshape(n, -1, 3)[:, lsps_ref.NYU_EVAL_JOINTS]
    pr = pose.reshape(n, -1, 3)[:, lsps_ref.NYU_EVAL_JOINTS]
    com = b['cb'].reshape(n, 1, 3)
    pr3d, gt3d = pr * (cube[0] / 2.) + com, gt * (cube[0] / 2.) + com
    err = np.sqrt(np.square(gt3d - pr3d).sum(axis=2))
    rel = np.abs(pose - ref['pose']).max() / np.abs(ref['pose']).max()
    print("bf16 joint read-out: rel err of the joints", rel)
    assert 1e-6 < rel <= 1e-2
    assert (np.argmax(err, axis=1) == ref['worst_joint']).all()
    assert int((np.nanmax(err, axis=1) <= 40).sum()) == ref['frames_within_40']
    assert abs(np.nanmean(np.nanmean(err, axis=1)) - ref['mean_err']) <= 1e-2 * ref['mean_err']


def test_full_batch_properties():
    """Size-independent checks at BASELINE's full size (bs=128 per domain, ch=64), where the oracle is
    too slow to run: per-sample independence (a sample's output does not depend on its batch-mates),
    linearity of conv in its input, and tanh range."""
    A = _adapter()
    hp = cases.hp_for('full')
    sds = cases.make_weights(hp, lsps_ref)
    tr = A.make_trainer(hp, sds)
    A.set_train(tr, False)
    n = 128
    b = cases.make_inputs(n)
    with torch.no_grad():
        xa, xb = A.T(b['xa']), A.T(b['xb'])
        big = tr.gen(xa, xb)
        small = tr.gen(xa[5:7].contiguous(), xb[5:7].contiguous())
        for t_big, t_small in zip(big[:4], small[:4]):
            assert float((t_big[5:7] - t_small).abs().max()) <= 1e-5
            assert float(t_big.abs().max()) <= 1.0
        post_big = tr.dis.regress_b(xb)[1]
        post_small = tr.dis.regress_b(xb[40:44].contiguous())[1]
        assert float((post_big[40:44] - post_small).abs().max()) <= 1e-4 * float(post_big.abs().max())


def test_full_size_step_gradients_equal_the_mean_of_chunk_gradients():
    _chunk_mean_property(cases.hp_for('full'), 128, 8, 108, None, 5e-3, 5e-4)


def test_config5_full_size_bf16_step_gradients_equal_the_mean_of_chunk_gradients():
    """The same property at BASELINE config 5's size and mode: exps/nicvl.yaml, bf16 math mode, 256 samples per domain — the C8
    kernels at their bench grids (tiles of whole images on the discriminator trunk's small maps with 1536 images, 512-image
    residual convs, one round of weight-gradient workgroups) against thirty-two 8-sample chunks through the same kernels at
    small grids.  Per sample the bf16 kernels do the same arithmetic at any batch size; what differs is tile mapping, ragged
    last tiles and the f32 summation order of the weight gradients and bias-gradient partial sums."""
    from lsps_amd import ops
    ops.set_math_mode('bf16')
    try:
        _chunk_mean_property(cases.load_hp('nicvl'), 256, 8, 48, 'bf16', 1e-4, 2e-5)      # measured: worst 4.6e-6, 90 % below 9.3e-7
    finally:
        ops.set_math_mode('f32')


def _chunk_mean_property(hp, n, ch, label_dim, mode, tol_worst, tol_90):
    """BASELINE's full size (bs=128 per domain, full width), backward included: every loss of dis_update / gen_update is a
    batch mean and InstanceNorm has no batch statistics, so the gradients of the 128-sample step are the MEAN of the gradients
    of its sixteen 8-sample chunks (same weights, the chunks' rows of the same noise), and so are the loss scalars.  The full
    step runs the default dispatch of the bench (F(4x4,3x3) kernels with the fused norms, XCD-aware mappings, the stride-2
    kernels at full grids, the batch-innermost trunk); the chunks run the small-grid paths that the reference's golden
    vectors pin at N = 2 ... 8: a wrong dgrad / wgrad / norm-backward at full size shows as a broken mean."""
    A = _adapter()
    sds = cases.make_weights(hp, lsps_ref)
    b = cases.make_inputs(n, label_dim=label_dim)
    lat2, lat1 = cases.latent_shape(hp, 2 * n), cases.latent_shape(hp, n)
    nz_d = cases.noise(lat2, 11)
    nz_g = (cases.noise(lat2, 21), cases.noise(lat1, 31), cases.noise(lat1, 41))

    def rows2(z, j):                     # rows of chunk j of a [2n] tensor laid out (domain a | domain b)
        return np.concatenate((z[j * ch:(j + 1) * ch], z[n + j * ch:n + (j + 1) * ch]), 0)

    def reload(tr):
        for net in ('gen', 'dis', 'vae', 'map'):
            getattr(tr, net).load_state_dict({k: torch.as_tensor(v) for k, v in sds[net].items()})

    tr = A.make_trainer(hp, sds)
    A.set_train(tr, True)
    A.dis_update(tr, b, hp, nz_d)
    full = dict(dis=A.grads(tr, 'dis'), dis_s=A.scalars(tr))
    reload(tr)
    A.gen_update(tr, b, hp, nz_g)
    full.update(gen=A.grads(tr, 'gen'), gen_s=A.scalars(tr))
    acc = dict(dis=None, gen=None, dis_s=None, gen_s=None)

    def add(key, val):
        if acc[key] is None:
            acc[key] = {k: np.asarray(v, np.float64).copy() for k, v in val.items() if v is not None}
        else:
            for k in acc[key]:
                acc[key][k] += val[k]

    for j in range(n // ch):
        bj = {k: v[j * ch:(j + 1) * ch] for k, v in b.items()}
        reload(tr)
        A.dis_update(tr, bj, hp, rows2(nz_d, j))
        add('dis', A.grads(tr, 'dis'))
        add('dis_s', A.scalars(tr))
        reload(tr)
        A.gen_update(tr, bj, hp, (rows2(nz_g[0], j), nz_g[1][j * ch:(j + 1) * ch], nz_g[2][j * ch:(j + 1) * ch]))
        add('gen', A.grads(tr, 'gen'))
        add('gen_s', A.scalars(tr))
    k_chunks = float(n // ch)
    worst = {}
    errs = []
    for net in ('dis', 'gen'):
        for name, g in full[net].items():
            if g is None or name not in acc[net]:
                continue
            mean = acc[net][name] / k_chunks
            scale = max(float(np.abs(mean).max()), 1e-12)
            err = float(np.abs(np.asarray(g, np.float64) - mean).max()) / scale
            worst[net] = max(worst.get(net, 0.0), err)
            errs.append((err, net, name))
    # Tolerance: the two sides differ by fp32 round-off of different algorithms (F(4x4,3x3) vs direct / F(2x2,3x3): ~1e-5 per
    # conv, 28 convs deep) and the gradients are sums of signed terms that cancel (the feature-matching L1 loss feeds sign()
    # into the last trunk layers: measured 1.5e-3 of the tensor's abs-max there, median over all tensors 1e-4).  A wrong
    # kernel is off by O(0.1 .. 1).  The golden step-gradient bound of tests/golden/cases.py is 2e-2.
    errs.sort(reverse=True)
    print("chunk-mean property (%s): worst %s, 90th percentile %s" % (mode or 'f32', errs[0], errs[len(errs) // 10]))
    assert errs[0][0] <= tol_worst, errs[:5]
    assert errs[len(errs) // 10][0] <= tol_90, errs[len(errs) // 10]          # 90 % of the tensors
    for key, names in (('dis_s', ('dis_loss', 'dis_ad_loss', 'dis_feat_loss', 'dis_true_acc', 'dis_fake_acc')),
                       ('gen_s', ('gen_total_loss',))):
        for name in names:
            if name in full[key]:
                m = acc[key][name] / k_chunks
                assert abs(float(full[key][name]) - float(m)) <= 1e-3 * max(abs(float(m)), 1e-6), (name, full[key][name], m)
    assert worst['dis'] > 0.0 and worst['gen'] > 0.0          # different kernels really ran (not the same bits twice)


def test_full_size_post_update_mode3_is_chunk_regression_mean_plus_first4_feature_term():
    """The literal `estimate3` step at BASELINE's full size (post_update(mode=3), bs=128 per domain, full width), backward
    included.  Its loss is reg_w * mean_n ||regress_a(x_n) - code_n||^2 (a batch mean) + feature_w_reg * the feature term of
    the FIRST FOUR samples per domain (batch-independent; lsps_trainer.py:238).  So with sixteen 8-sample chunks:
        grad(full, mode 3) = mean_j grad(chunk j, mode 0) + [grad(chunk 0, mode 3) - grad(chunk 0, mode 0)]
    (mode 0 = the same regression term alone, :227-230).  The full step runs the bench's dispatch (batch-innermost trunk at
    N=128, side stream), the chunks the small-batch paths the golden vectors pin at N=8."""
    A = _adapter()
    hp = cases.hp_for('full')
    sds = cases.make_weights(hp, lsps_ref)
    n, ch = 128, 8
    b = cases.make_inputs(n)
    zd = hp['vae']['z_dim']
    nz_gen = cases.noise(cases.latent_shape(hp, 8), 51)
    nz_va, nz_vb = cases.noise((n, zd), 52, 0.05), cases.noise((n, zd), 53, 0.05)

    def reload(tr):
        for net in ('gen', 'dis', 'vae', 'map'):
            getattr(tr, net).load_state_dict({k: torch.as_tensor(v) for k, v in sds[net].items()})

    def grads64(tr):
        return {k: np.asarray(v, np.float64) for k, v in A.grads(tr, 'dis').items() if v is not None}

    tr = A.make_trainer(hp, sds)
    A.set_train(tr, True)
    A.post_update(tr, b, 3, hp, nz_gen, nz_va, nz_vb)
    full, full_s = grads64(tr), A.scalars(tr)
    acc, reg_acc = None, 0.0
    for j in range(n // ch):
        bj = {k: v[j * ch:(j + 1) * ch] for k, v in b.items()}
        reload(tr)
        A.post_update(tr, bj, 0, hp, nz_gen, nz_va[j * ch:(j + 1) * ch], nz_vb[j * ch:(j + 1) * ch])
        g = grads64(tr)
        reg_acc += float(A.scalars(tr)['dis_reg_loss'])
        if j == 0:
            g0 = g
        acc = g if acc is None else {k: acc[k] + g[k] for k in acc}
    reload(tr)
    b0 = {k: v[0:ch] for k, v in b.items()}
    A.post_update(tr, b0, 3, hp, nz_gen, nz_va[0:ch], nz_vb[0:ch])
    g3 = grads64(tr)
    errs = []
    for name, g in full.items():
        if name not in acc:
            continue
        expect = acc[name] / float(n // ch) + (g3[name] - g0[name])
        scale = max(float(np.abs(expect).max()), 1e-12)
        errs.append((float(np.abs(g - expect).max()) / scale, name))
    errs.sort(reverse=True)
    assert errs and errs[0][0] <= 5e-3, errs[:5]
    assert errs[len(errs) // 10][0] <= 5e-4, errs[len(errs) // 10]
    assert abs(float(full_s['dis_reg_loss']) - reg_acc / (n // ch)) <= 1e-3 * abs(reg_acc / (n // ch))
    assert errs[0][0] > 0.0


def test_update_steps_are_bitwise_deterministic():
    """No atomics anywhere on the path (split reductions are two-stage): the same step from the same state
    gives bit-identical losses and weights."""
    A = _adapter()
    hp = cases.hp_for('tiny')
    sds = cases.make_weights(hp, lsps_ref)
    b = cases.make_inputs(4)
    lat2, lat1 = cases.latent_shape(hp, 8), cases.latent_shape(hp, 4)
    outs = []
    for _ in range(2):
        tr = A.make_trainer(hp, sds)
        A.set_train(tr, True)
        A.dis_update(tr, b, hp, cases.noise(lat2, 1))
        A.gen_update(tr, b, hp, (cases.noise(lat2, 2), cases.noise(lat1, 3), cases.noise(lat1, 4)))
        outs.append((A.scalars(tr), A.params(tr, 'gen'), A.params(tr, 'dis')))
    assert outs[0][0] == outs[1][0]
    for net in (1, 2):
        for k in outs[0][net]:
            assert np.array_equal(outs[0][net][k], outs[1][net][k]), k


def test_workspace_grows_inside_a_graph_capture():
    """The scratch buffer of a stream may have to grow while that stream is being captured (full-width nets at bs=8: the
    reduction-split partials of the residual convs need more than the first conv of the step): no synchronisation is allowed
    there, the outgrown buffer must stay alive for the kernels captured so far, and the replay must still be right."""
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from lsps_amd import _lib, ops
    dev = torch.device('cuda', 0)
    x = torch.randn(2, 8, 16, 32, device=dev)
    w1 = torch.randn(8, 8, 3, 3, device=dev) * 0.1
    big = torch.randn(2, 64, 32, 32, device=dev)
    w2 = torch.randn(64, 64, 3, 3, device=dev) * 0.1
    ref1, ref2 = ops.conv2d(x, w1, None, 1, 1), ops.conv2d(big, w2, None, 1, 1)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        cap = torch.cuda.current_stream(dev).cuda_stream
        p0, n0 = _lib.workspace(1 << 20, dev)
        y1 = ops.conv2d(x, w1, None, 1, 1)
        p1, n1 = _lib.workspace(n0 + (32 << 20), dev)    # grows during capture
        assert n1 >= n0 + (32 << 20) and p1 != p0
        y2 = ops.conv2d(big, w2, None, 1, 1)
    g.replay()
    torch.cuda.synchronize()
    assert torch.equal(y1, ref1) and torch.equal(y2, ref2)
    assert any(k[2] == cap for k in _lib._workspaces)


def test_hip_graph_replay_matches_eager_bitwise():
    """`use_graphs(True)`: per call signature the first call is eager, the second is captured into a hipGraph, later ones
    replay it.  Five rounds of dis_update + gen_update + post_update(mode 3) on DIFFERENT inputs / injected noise each
    round (so a replay that read stale static buffers would show) give bit-identical loss scalars and weights to the eager
    trainer, and the graphs really were replayed."""
    A = _adapter()
    hp = cases.hp_for('tiny')
    sds = cases.make_weights(hp, lsps_ref)
    lat2, lat1, lat4 = cases.latent_shape(hp, 8), cases.latent_shape(hp, 4), cases.latent_shape(hp, 8)
    zd = hp['vae']['z_dim']
    outs = []
    for graphed in (False, True):
        tr = A.make_trainer(hp, sds)
        tr.use_graphs(graphed)
        A.set_train(tr, True)
        trace = []
        for rnd in range(5):
            if graphed and rnd == 3:            # graphs off and on again: the old graphs and their memory pool are gone,
                tr.use_graphs(False)            # round 3 runs eagerly, round 4 captures into a fresh pool
                tr.use_graphs(True)
            b = cases.make_inputs(4)
            b = {k: (v * (1.0 - 0.1 * rnd)).astype(v.dtype) if k in ('xa', 'xb') else v for k, v in b.items()}
            A.dis_update(tr, b, hp, cases.noise(lat2, 10 + rnd))
            A.gen_update(tr, b, hp, (cases.noise(lat2, 20 + rnd), cases.noise(lat1, 30 + rnd), cases.noise(lat1, 40 + rnd)))
            A.post_update(tr, b, 3, hp, cases.noise(lat4, 50 + rnd), cases.noise((4, zd), 60 + rnd, 0.05),
                          cases.noise((4, zd), 70 + rnd, 0.05))
            trace.append(A.scalars(tr))
        if graphed:
            assert len(tr._graphs) == 3, sorted(k[0] for k in tr._graphs)
        outs.append((trace, A.params(tr, 'gen'), A.params(tr, 'dis')))
    assert outs[0][0] == outs[1][0], (outs[0][0], outs[1][0])
    for net in (1, 2):
        for k in outs[0][net]:
            assert np.array_equal(outs[0][net][k], outs[1][net][k]), k


@pytest.mark.parametrize("mode", [3, 4])
def test_post_update_overlapped_branches_match_serial_bitwise(mode, monkeypatch):
    """Estimate modes: the feature branch (generator on the first 4 samples + dis.feats) runs on a second HIP stream beside
    the regression branch.  Two steps with the overlap, without it (LSPS_NO_OVERLAP=1) and with the overlap inside a
    hipGraph give bit-identical losses and discriminator weights."""
    A = _adapter()
    hp = cases.hp_for('tiny')
    sds = cases.make_weights(hp, lsps_ref)
    b = cases.make_inputs(8)
    lat, zd = cases.latent_shape(hp, 8), hp['vae']['z_dim']
    outs = []
    for variant in ('overlap', 'serial', 'graph'):
        from lsps_amd import options
        env = {'LSPS_EST_MERGE': '0'}           # the two-pass schedule of round 4 (round 5's default merges the passes: next test)
        if variant == 'serial':
            env['LSPS_NO_OVERLAP'] = '1'
        monkeypatch.setattr(options, '_current', options.from_env(env))
        tr = A.make_trainer(hp, sds)
        tr.use_graphs(variant == 'graph')
        A.set_train(tr, True)
        trace = []
        for rnd in range(3):
            A.post_update(tr, b, mode, hp, cases.noise(lat, 80 + rnd), cases.noise((8, zd), 81 + rnd, 0.05),
                          cases.noise((8, zd), 82 + rnd, 0.05))
            trace.append(A.scalars(tr))
        assert (tr._side is not None) == (variant != 'serial')
        outs.append((trace, A.params(tr, 'dis')))
    for o in outs[1:]:
        assert o[0] == outs[0][0], (o[0], outs[0][0])
        for k in outs[0][1]:
            assert np.array_equal(o[1][k], outs[0][1][k]), k


@pytest.mark.parametrize("config", ["tiny", "full"])
def test_shared_encoder_pass_gives_the_same_iteration(config, monkeypatch):
    """Opt-in `options.share_encoder`: the encoder half of `gen(images_a, images_b)` runs once per pretrain iteration (dis_update keeps
    its tape, gen_update continues from it; the reference runs it twice with the same images and weights, lsps_trainer.py:86,145).
    Two iterations with and without: the first gen_update's loss scalars are bit-identical, everything else agrees to round-off."""
    A = _adapter()
    from lsps_amd import options
    hp = cases.hp_for(config)
    sds = cases.make_weights(hp, lsps_ref)
    n = 2 if config == 'full' else 4
    b = cases.make_inputs(n)
    lat2, lat1 = cases.latent_shape(hp, 2 * n), cases.latent_shape(hp, n)
    res = []
    for share in ('0', '1'):
        monkeypatch.setattr(options, '_current', options.from_env({'LSPS_SHARE_ENCODER': share}))
        tr = A.make_trainer(hp, sds)
        A.set_train(tr, True)
        dev = {k: A.T(v) for k, v in b.items()}                      # the SAME device tensors go to both calls, as in depth_train.py
        trace, outs = [], None
        for it in range(2):
            nz = [A.T(cases.noise(lat2, 300 + it)), A.T(cases.noise(lat2, 310 + it)), A.T(cases.noise(lat1, 320 + it)),
                  A.T(cases.noise(lat1, 330 + it))]
            tr.dis_update(dev['xa'], dev['la'], dev['xb'], dev['lb'], dev['ca'], dev['cb'], hp, noise=nz[0])
            assert (tr._enc_shared_pass is not None) == (share == '1')
            outs = tr.gen_update(dev['xa'], dev['la'], dev['xb'], dev['lb'], hp, noise=(nz[1], nz[2], nz[3]))
            assert tr._enc_shared_pass is None
            trace.append(A.scalars(tr))
        res.append((trace, [A.N(o) for o in outs[:6]], A.params(tr, 'gen'), A.params(tr, 'dis')))
    (t0, o0, g0, d0), (t1, o1, g1, d1) = res
    # iteration 0: gen_update is bit-identical (its encoder pass IS the shared one: same kernels on the same weights); dis_update's
    # own numbers may move by round-off where its generator pass ran the differentiated kernels instead of the no-grad ones (full
    # width at this batch), and from there on the two runs are two round-off-different trainings: iteration 1 to 1e-4
    for k in t0[0]:
        if k.startswith('gen_'):
            assert t0[0][k] == t1[0][k], (k, t0[0][k], t1[0][k])
    for a, b_ in zip(t0, t1):
        for k in a:
            assert abs(float(a[k]) - float(b_[k])) <= 1e-4 * max(1.0, abs(float(a[k]))), (k, a[k], b_[k])
    for a, b_ in zip(o0, o1):
        assert float(np.abs(a - b_).max()) <= 1e-4 * float(np.abs(a).max())
    for p0, p1 in ((g0, g1), (d0, d1)):
        for k in p0:
            assert float(np.abs(p0[k] - p1[k]).max()) <= 6.5e-4, k   # two Adam steps of lr 1e-4 behind round-off-different gradients


@pytest.mark.parametrize("mode", [3, 4])
def test_post_update_merged_discriminator_pass_equals_the_two_pass_schedule(mode, monkeypatch):
    """Round 5: the estimate modes run `dis.regress_*` (whole batch) and `dis.feats` (16 generator outputs) as ONE pass of the
    discriminator (SharedDis.regress_feats).  Per sample the arithmetic is that of the separate calls; what changes is the
    summation order of the weight gradients.  Two steps of the merged schedule (eager and replayed from a hipGraph: bitwise
    equal) against the two-pass schedule: losses to 1e-5, discriminator weights after the Adam steps to the golden rule."""
    A = _adapter()
    from lsps_amd import options
    hp = cases.hp_for('tiny')
    sds = cases.make_weights(hp, lsps_ref)
    b = cases.make_inputs(8)
    lat, zd = cases.latent_shape(hp, 8), hp['vae']['z_dim']
    outs = {}
    for variant in ('two_pass', 'merged', 'merged_graph'):
        monkeypatch.setattr(options, '_current', options.from_env({'LSPS_EST_MERGE': '0' if variant == 'two_pass' else '1'}))
        tr = A.make_trainer(hp, sds)
        tr.use_graphs(variant == 'merged_graph')
        A.set_train(tr, True)
        trace = []
        for rnd in range(3):
            A.post_update(tr, b, mode, hp, cases.noise(lat, 80 + rnd), cases.noise((8, zd), 81 + rnd, 0.05),
                          cases.noise((8, zd), 82 + rnd, 0.05))
            trace.append(A.scalars(tr))
        outs[variant] = (trace, A.params(tr, 'dis'))
    assert outs['merged'][0] == outs['merged_graph'][0]
    for k in outs['merged'][1]:
        assert np.array_equal(outs['merged'][1][k], outs['merged_graph'][1][k]), k
    for t_m, t_2 in zip(outs['merged'][0], outs['two_pass'][0]):
        for k in t_m:
            assert abs(float(t_m[k]) - float(t_2[k])) <= 1e-5 * max(1.0, abs(float(t_2[k]))), (k, t_m[k], t_2[k])
    for k in outs['merged'][1]:
        a, b2 = outs['merged'][1][k], outs['two_pass'][1][k]
        d = np.abs(a.astype(np.float64) - b2.astype(np.float64))
        # three Adam steps of lr 1e-4: a weight whose gradient is round-off around 0 may step the other way
        assert float(d.max()) <= 6.5e-4 and float((d > 1e-6).mean()) <= 0.03, (k, float(d.max()), float((d > 1e-6).mean()))


@pytest.mark.parametrize("graphs", [False, True])
def test_frozen_generator_packs_survive_steps_and_follow_weight_changes(graphs, monkeypatch):
    """The estimate modes never step the generator (lsps_trainer.py:220-262), so its packed weight panels are kept from one
    post_update to the next (lsps_pack_cache_frozen).  Bitwise against a run that re-packs in every step, over a sequence that
    also CHANGES the generator between estimate steps three ways — a gen_update (Adam kernel: arena generation), a
    load_state_dict (torch version counters), the order post / post / gen_update / post / load / post — eager and from hipGraphs
    (a captured post_update holds no generator pack launches: it must be re-captured once the weights moved)."""
    A = _adapter()
    hp = cases.hp_for('tiny')
    sds = cases.make_weights(hp, lsps_ref)
    sds2 = cases.make_weights(hp, lsps_ref)
    sds2['gen'] = {k: (v * 1.25).astype(v.dtype) for k, v in sds2['gen'].items()}
    b = cases.make_inputs(8)
    lat, lat1, zd = cases.latent_shape(hp, 8), cases.latent_shape(hp, 8), hp['vae']['z_dim']
    outs = []
    for frozen in (True, False):
        from lsps_amd import options
        monkeypatch.setattr(options, '_current', options.from_env({} if frozen else {'LSPS_NO_FROZEN_PACKS': '1'}))
        if frozen:
            # a DECOY first: another trainer with other generator weights fills the frozen table, then goes away — the caching
            # allocator hands its arena's addresses to the next trainer, whose generation / version counts are the same
            decoy = A.make_trainer(hp, sds2)
            A.set_train(decoy, True)
            for rnd in range(3):
                A.post_update(decoy, b, 3, hp, cases.noise(lat, 70 + rnd), cases.noise((8, zd), 71 + rnd, 0.05),
                              cases.noise((8, zd), 72 + rnd, 0.05))
            decoy_ptr = decoy.gen_opt.arena.flat_p.data_ptr()
            del decoy
        tr = A.make_trainer(hp, sds)
        if frozen and not graphs:
            print("arena address reused by the next trainer:", tr.gen_opt.arena.flat_p.data_ptr() == decoy_ptr)
        tr.use_graphs(graphs)
        A.set_train(tr, True)
        trace, epochs = [], []

        def post(rnd):
            A.post_update(tr, b, 3, hp, cases.noise(lat, 90 + rnd), cases.noise((8, zd), 91 + rnd, 0.05),
                          cases.noise((8, zd), 92 + rnd, 0.05))
            trace.append(A.scalars(tr))
            epochs.append(tr.gen_opt.arena.epoch())
        post(0)
        post(1)
        post(2)
        A.gen_update(tr, b, hp, (cases.noise(cases.latent_shape(hp, 16), 95), cases.noise(lat1, 96), cases.noise(lat1, 97)))
        post(3)
        post(4)
        tr.gen.load_state_dict({k: torch.as_tensor(v) for k, v in sds2['gen'].items()})
        post(5)
        post(6)
        post(7)
        post(8)
        assert epochs[0] == epochs[1] == epochs[2] and epochs[3] == epochs[4] and epochs[5] == epochs[8]
        assert len(set(epochs)) == 3, epochs                      # every way of changing the weights moved the counter
        if graphs:
            # a graph captured while the panels were frozen carries the generator epoch it is valid for; those of older epochs
            # are dropped, the one of the current epoch exists (and was replayed by the last call)
            # (the tag is (epoch, count of re-targetings of the library's process-wide frozen table): ops.frozen_resets)
            tags = [k[-1][1] for k in tr._graphs if k[0] == 'post_update']
            eps = [None if t is None else t[0] for t in tags]
            assert all(e in (None, epochs[-1]) for e in eps), (eps, epochs)
            assert (epochs[-1] in eps) == frozen, (eps, frozen)
        outs.append((trace, A.params(tr, 'dis'), A.params(tr, 'gen')))
    assert outs[0][0] == outs[1][0], (outs[0][0], outs[1][0])
    for net in (1, 2):
        for k in outs[0][net]:
            assert np.array_equal(outs[0][net][k], outs[1][net][k]), k


def test_two_graphed_trainers_alternating_post_update_keep_their_own_frozen_panels():
    """the library's frozen table is one per process.  Two trainers with DIFFERENT generator weights, both replaying
    their estimate steps from hipGraphs, alternate post_update in one process: each trainer's results equal, bitwise, those of
    the same trainer stepping alone in eager mode (every trainer owns its panel buffer, and a graph captured while panels were
    frozen is dropped as soon as the table has been re-targeted by anybody)."""
    A = _adapter()
    hp = cases.hp_for('tiny')
    sds = [cases.make_weights(hp, lsps_ref), cases.make_weights(hp, lsps_ref)]
    sds[1]['gen'] = {k: (v * 1.25).astype(v.dtype) for k, v in sds[1]['gen'].items()}
    b = cases.make_inputs(8)
    lat, zd = cases.latent_shape(hp, 8), hp['vae']['z_dim']

    def post(tr, rnd):
        A.post_update(tr, b, 3, hp, cases.noise(lat, 60 + rnd), cases.noise((8, zd), 61 + rnd, 0.05), cases.noise((8, zd), 62 + rnd, 0.05))
        return A.scalars(tr)
    alone = []
    for sd in sds:
        tr = A.make_trainer(hp, sd)
        A.set_train(tr, True)
        alone.append(([post(tr, r) for r in range(6)], A.params(tr, 'dis')))
        del tr
    trs = [A.make_trainer(hp, sd) for sd in sds]
    for tr in trs:
        tr.use_graphs(True)
        A.set_train(tr, True)
    # two steps each first (the panels freeze once two consecutive post_update calls saw the same generator epoch), then strictly
    # alternating, then a run of one trainer (which captures and replays a frozen graph) before the other comes back
    order = [0, 0, 1, 1, 0, 1, 0, 1, 0, 0, 1, 1]
    traces, count = [[], []], [0, 0]
    for i in order:
        traces[i].append(post(trs[i], count[i]))
        count[i] += 1
    for i in (0, 1):
        assert traces[i] == alone[i][0], (i, traces[i], alone[i][0])
        got = A.params(trs[i], 'dis')
        for k in got:
            assert np.array_equal(got[k], alone[i][1][k]), (i, k)
    assert traces[0] != traces[1]


def test_two_trainers_in_different_math_modes_alternate_in_one_process():
    """The library's math / Winograd modes are process-wide; `LSPSTrainer.set_modes` installs a trainer's own around each of its
    update methods (two trainers in different modes could not coexist).  An f32 and a bf16 trainer stepped
    alternately give, bitwise, what each gives alone under the corresponding process-wide mode — and the process-wide mode is
    what it was afterwards."""
    A = _adapter()
    from lsps_amd import ops
    hp = cases.hp_for('full')
    sds = cases.make_weights(hp, lsps_ref)
    b = cases.make_inputs(2)
    lat2, lat1 = cases.latent_shape(hp, 4), cases.latent_shape(hp, 2)

    def steps(tr, rounds):
        out = []
        for r in rounds:
            A.dis_update(tr, b, hp, cases.noise(lat2, 500 + r))
            A.gen_update(tr, b, hp, (cases.noise(lat2, 510 + r), cases.noise(lat1, 520 + r), cases.noise(lat1, 530 + r)))
            out.append(A.scalars(tr))
        return out
    alone = {}
    for mode in ('f32', 'bf16'):
        ops.set_math_mode(mode)
        try:
            tr = A.make_trainer(hp, sds)
            A.set_train(tr, True)
            alone[mode] = (steps(tr, (0, 1)), A.params(tr, 'gen'))
        finally:
            ops.set_math_mode('f32')
    t32, t16 = A.make_trainer(hp, sds), A.make_trainer(hp, sds).set_modes(math_mode='bf16')
    A.set_train(t32, True)
    A.set_train(t16, True)
    tr32, tr16 = [], []
    for r in (0, 1):
        tr32 += steps(t32, (r,))
        assert ops.get_math_mode() == 'f32'
        tr16 += steps(t16, (r,))
        assert ops.get_math_mode() == 'f32'
    assert alone['f32'][0] != alone['bf16'][0], "the bf16 mode did not engage"
    for mode, trace, tr in (('f32', tr32, t32), ('bf16', tr16, t16)):
        assert trace == alone[mode][0], mode
        got = A.params(tr, 'gen')
        for k, v in alone[mode][1].items():
            assert np.array_equal(got[k], v), (mode, k)


@pytest.mark.parametrize("n", [1, 2, 3, 5])
def test_ragged_batches_against_oracle(n):
    """Edge cases the reference's code paths have: batch smaller than the [0:4] slice of post_update
    (lsps_trainer.py:238), n == 1 (.squeeze() drops the batch axis, lsps_nets.py:139), odd batches."""
    A = _adapter()
    O = cases.NativeAdapter(lsps_ref, 'cpu', trainer_kwargs=dict(literal=False))
    hp = cases.hp_for('tiny')
    sds = cases.make_weights(hp, lsps_ref)
    b = cases.make_inputs(n)
    k = min(n, 4)
    latp = cases.latent_shape(hp, 2 * k)
    zd = hp['vae']['z_dim']
    res = []
    for Ad in (O, A):
        tr = Ad.make_trainer(hp, sds)
        Ad.set_train(tr, True)
        mode = 3 if n != 1 else 0          # n == 1: dis.feats' split-by-4 needs >= 1 sample per domain pair
        Ad.post_update(tr, b, mode, hp, cases.noise(latp, 5), cases.noise((n, zd), 6, 0.05), cases.noise((n, zd), 7, 0.05))
        Ad.set_train(tr, False)
        post = Ad.dis_regress(tr, 'b', b['xb'])[1]
        res.append((Ad.scalars(tr), post, Ad.vae_decode(tr, post)))
    for k_ in res[0][0]:
        assert abs(res[0][0][k_] - res[1][0][k_]) <= 1e-3 * max(1e-6, abs(res[0][0][k_])), k_
    assert res[0][1].shape == res[1][1].shape                     # [20] at n == 1, [n, 20] otherwise
    assert np.abs(res[0][1] - res[1][1]).max() <= 1e-3 * np.abs(res[0][1]).max()
    assert np.abs(res[0][2] - res[1][2]).max() <= 1e-3 * np.abs(res[0][2]).max()


def test_bf16_math_mode_end_to_end(golden):
    """BASELINE config 5 (bf16 MFMA conv path): full-width nets with the residual convs on bf16 MFMA operands
    (f32 accumulation, f32 InstanceNorm statistics, f32 master weights / Adam) against the SAME f32 golden
    vectors of the reference.  bf16 carries 8 mantissa bits: forward tensors within 3e-2 of their abs-max,
    loss scalars within 5 %."""
    A = _adapter()
    from lsps_amd import ops
    ops.set_math_mode('bf16')
    try:
        R = cases.run_module_cases(A, 'full', lsps_ref)
        hp = cases.hp_for('full')
        sds = cases.make_weights(hp, lsps_ref)
        tr = A.make_trainer(hp, sds)
        A.set_train(tr, True)
        b = cases.make_inputs(2)
        lat2, lat1 = cases.latent_shape(hp, 4), cases.latent_shape(hp, 2)
        A.dis_update(tr, b, hp, cases.noise(lat2, 1000))
        A.gen_update(tr, b, hp, (cases.noise(lat2, 2000), cases.noise(lat1, 3000), cases.noise(lat1, 4000)))
        scal = A.scalars(tr)
    finally:
        ops.set_math_mode('f32')
    g = golden('full')
    gm = {k: v for k, v in g.items() if k.split('/')[0] in R}
    bad, worst = cases.compare(R, gm, 3e-2)
    print("bf16 worst forward rel err", worst)
    assert not bad, "worst=%g first failures: %s" % (worst, bad[:6])
    assert worst > 1e-5, "bf16 mode did not engage"
    for k, v in scal.items():
        ref = float(g['pretrain.it0.gen_update.scalars/%s/full' % k])
        assert abs(v - ref) <= 5e-2 * max(abs(ref), 1e-3), (k, v, ref)


@pytest.mark.parametrize("n", [2, 3, 5])
def test_icvl_config_bf16_against_oracle(n):
    """BASELINE config 5 itself: exps/nicvl.yaml (vae.input_dim = 48), bf16 MFMA conv path, against the f32 CPU
    oracle on the same seeded inputs (no golden vectors exist for ICVL: the oracle is the checker here).  n = 3, 5: ragged
    batches (the C8 kernels' tiles of whole images / image octets are partly empty)."""
    A = _adapter()
    from lsps_amd import ops
    hp = cases.load_hp('nicvl')
    assert hp['vae']['input_dim'] == 48
    sds = cases.make_weights(hp, lsps_ref)
    b = cases.make_inputs(n, label_dim=48)
    lat2, lat1 = cases.latent_shape(hp, 2 * n), cases.latent_shape(hp, n)
    nz = (cases.noise(lat2, 21), cases.noise(lat2, 22), cases.noise(lat1, 23), cases.noise(lat1, 24))
    zd = hp['vae']['z_dim']
    res = []
    for Ad, mode in ((cases.NativeAdapter(lsps_ref, 'cpu', trainer_kwargs=dict(literal=False)), None), (A, 'bf16')):
        if mode:
            ops.set_math_mode(mode)
        try:
            tr = Ad.make_trainer(hp, sds)
            Ad.set_train(tr, True)
            Ad.dis_update(tr, b, hp, nz[0])
            outs = Ad.gen_update(tr, b, hp, nz[1:])
            Ad.post_update(tr, b, 3, hp, cases.noise(cases.latent_shape(hp, 2 * min(n, 4)), 25), cases.noise((n, zd), 26, 0.05),
                           cases.noise((n, zd), 27, 0.05))
            res.append((Ad.scalars(tr), outs[0], outs[4]))
        finally:
            if mode:
                ops.set_math_mode('f32')
    (s_ref, xaa_ref, xaba_ref), (s_hip, xaa, xaba) = res
    for k in s_ref:
        assert abs(s_ref[k] - s_hip[k]) <= 5e-2 * max(abs(s_ref[k]), 1e-3), (k, s_ref[k], s_hip[k])
    assert np.abs(xaa - xaa_ref).max() <= 3e-2 and np.abs(xaba - xaba_ref).max() <= 3e-2


def test_resblock_dropout_against_reference():
    """`res_dropout_ratio` > 0: the HIP residual block (IN kernel + lsps_mul_add) with the recorded keep mask against the
    reference's own LeakyINSResBlock(dropout=p), forward and backward; eval mode is the plain fused block; a generator
    built with the key runs and draws its own masks in training mode."""
    if not torch.cuda.is_available():
        pytest.skip("no HIP device")
    import os
    import yaml
    from lsps_amd import synth, trainers
    from lsps_amd.trainers.common_net import LeakyINSResBlock
    G = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'golden_dropout.npz'))
    d = cases.dropout_case_inputs()
    blk = LeakyINSResBlock(cases.DROP_CH, cases.DROP_CH, dropout=cases.DROP_P).cuda()
    assert len(blk.model) == 6                       # Dropout is child 5, as in the reference
    blk.load_state_dict({'model.0.weight': torch.as_tensor(d['w0']), 'model.0.bias': torch.as_tensor(d['b0']),
                         'model.3.weight': torch.as_tensor(d['w3']), 'model.3.bias': torch.as_tensor(d['b3'])})
    blk.train()
    x = torch.as_tensor(d['x']).cuda().requires_grad_(True)
    y = blk(x, drop_mask=torch.as_tensor(d['mask']).cuda())
    y.backward(torch.as_tensor(d['gy']).cuda())
    for name, got in (('y', y), ('dx', x.grad), ('dw0', blk.model[0].weight.grad), ('dw3', blk.model[3].weight.grad)):
        want = G['drop.train.' + name]
        err = np.abs(got.detach().cpu().numpy() - want).max() / max(1.0, np.abs(want).max())
        assert err <= 1e-3, (name, err)
    blk.eval()
    with torch.no_grad():
        ye = blk(torch.as_tensor(d['x']).cuda())
    assert np.abs(ye.cpu().numpy() - G['drop.eval.y']).max() <= 1e-3
    # own masks in training mode: about p of the branch is dropped, the skip connection is untouched
    blk.train()
    with torch.no_grad():
        xin = torch.as_tensor(d['x']).cuda()
        branch = blk(xin) - xin
    frac0 = float((branch == 0).float().mean())
    assert abs(frac0 - cases.DROP_P) < 0.02, frac0
    # the YAML key is honoured by the generator
    hp = yaml.safe_load(open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'exps',
                                          'nnyu.yaml')))['train']['hyperparameters']
    hp = synth.tiny_hyperparameters(hp)
    hp['gen']['res_dropout_ratio'] = 0.25
    gen = trainers.SharedResGen(hp['gen']).cuda()
    assert sum(1 for m in gen.modules() if isinstance(m, LeakyINSResBlock) and m.dropout == 0.25) == 14
    xa, _, _ = synth.make_batch(2, 3)
    out = gen(torch.as_tensor(xa).cuda(), torch.as_tensor(xa).cuda())
    assert all(bool(torch.isfinite(t).all()) for t in out)
