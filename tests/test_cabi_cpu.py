"""CPU-side checks of the drop-in boundary: the shared library loads and exports every symbol that
include/lsps_hip.h declares (no compute calls — there is no GPU here), the Python package mirrors
the reference's `trainers` namespace and state-dict keys, and the product path refuses to run
without the HIP device."""
import os
import re

import pytest
import torch
import yaml

import cases
from oracle import lsps_ref

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _header_functions():
    src = open(os.path.join(REPO, 'include', 'lsps_hip.h')).read()
    src = re.sub(r'/\*.*?\*/', '', src, flags=re.S)
    return sorted(set(re.findall(r'\b(lsps_[a-zA-Z0-9_]+)\s*\(', src)))


def test_library_exports_every_declared_symbol():
    from lsps_amd import _lib
    declared = _header_functions()
    assert len(declared) >= 20
    h = _lib.lib()
    for name in declared:
        assert hasattr(h, name), "liblsps_hip.so does not export %s" % name
    assert sorted(_lib.EXPORTS) == declared, "ctypes table and header disagree"
    assert h.lsps_version() == 1


def test_argument_validation_without_gpu():
    """Entry points reject bad arguments before touching the device."""
    from lsps_amd import _lib
    h = _lib.lib()
    assert h.lsps_conv2d_fwd(None, None, None, None, 1, 1, 8, 8, 1, 3, 3, 1, 1, 0, 0.0, None, 0, None) == -1
    assert b'null' in h.lsps_last_error()
    assert h.lsps_conv2d_workspace_bytes(2, 256, 32, 32, 256, 3, 3, 1, 1) > 0
    assert h.lsps_conv2d_workspace_bytes(2, 256, 32, 32, 256, 9, 9, 1, 1) == 0      # > 49 taps unsupported
    assert h.lsps_loss_workspace_bytes(10) > 0


def test_x3_launch_plan_fills_the_chip_on_the_deep_discriminator_layers():
    """The three-limb stride-2 kernels are persistent (one workgroup per CU) and walk (pixel tile, m tile, k range) units; with
    the XCD walk alone the estimate-mode batch (128 + 16 samples) left five of eight XCDs idle on `model_S.3`'s dgrad (three
    pixel tiles).  The plan (csrc/x3.hip: x3_plan) is computable without a GPU: deep layers split and take the linear walk,
    large-map layers keep one range and the XCD walk, and no plan starts more workgroups than CUs."""
    import ctypes
    from lsps_amd import _lib
    h = _lib.lib()

    def plan(tr, N, C, H, K):
        out = (ctypes.c_int * 4)()
        assert h.lsps_x3_conv3x3s2_plan(tr, N, C, H, H, K, out) == 0, h.lsps_last_error()
        return tuple(out)
    # model_S.3 (lsps_nets.py:119-121: 1024 -> 2048, 4x4 -> 2x2) at N = 144: dgrad has 3 pixel tiles x 16 m tiles
    ks, kper, linear, grid = plan(1, 144, 1024, 4, 2048)
    assert linear == 1 and ks >= 4 and ks * kper >= 128 and 200 <= grid <= 256
    ks, kper, linear, grid = plan(0, 144, 1024, 4, 2048)
    assert linear == 1 and ks >= 2 and 200 <= grid <= 256
    # the generator's down-sampling layer (64 -> 128 at 128x128, N = 256) and the 8-sample estimate-mode decoder: untouched
    assert plan(0, 256, 64, 128, 128)[:3] == (1, 4, 0)
    assert plan(1, 8, 128, 64, 256)[0] == 1            # ConvTranspose2d(256 -> 128) forward = dgrad of Conv2d(128 -> 256): 16.8 MB of output
    for N in (1, 5, 16, 40, 144, 256, 384, 768):
        for C, H, K in ((64, 64, 128), (128, 32, 256), (256, 16, 512), (512, 8, 1024), (1024, 4, 2048)):
            for tr in (0, 1):
                ks, kper, linear, grid = plan(tr, N, C, H, K)
                nk = (K if tr else C) // 16
                assert 1 <= ks <= 8 and (ks - 1) * kper < nk <= ks * kper and 1 <= grid <= 256, (tr, N, C, H, K, ks, kper, grid)
    assert h.lsps_x3_conv3x3s2_plan(0, 4, 60, 32, 32, 128, (ctypes.c_int * 4)()) == -1      # C % 16 != 0


def test_namespace_matches_reference_package():
    import lsps_amd.trainers as t
    # every public name of the reference's `from trainers import *` (probed list)
    for name in ('LSPSTrainer', 'SharedResGen', 'SharedResXGen', 'SharedDis', 'poseVAE', 'Mapping', 'Bias2d',
                 'GaussianNoiseLayer', 'GaussianSmoother', 'GaussianVAE', 'GaussianVAE2D', 'INSResBlock',
                 'LeakyINSResBlock', 'LeakyINSResNeXtBlock', 'LeakyReLUBNConv2d', 'LeakyReLUBNConvTranspose2d',
                 'LeakyReLUBNLinear', 'LeakyReLUBNNSConv2d', 'LeakyReLUBNNSConvTranspose2d', 'LeakyReLUBNNSResBlock',
                 'LeakyReLUConv2d', 'LeakyReLUConvTranspose2d', 'LeakyReLUINSConv2d', 'LeakyReLUINSConvTranspose2d',
                 'LeakyReLULinear', 'LeakyReLUResBlock', 'ReLUINSConv2d', 'ReLUINSConvTranspose2d',
                 'gaussian_weights_init', 'xavier_weights_init', 'get_model_list', 'Variable', 'torch', 'nn', 'os', 'np'):
        assert hasattr(t, name), name
    # the BatchNorm / ReLU variants construct (state-dict keys as in the reference) but, like everything else, only run on
    # HIP tensors: no torch fallback
    m = t.LeakyReLUBNConv2d(1, 2, 3, 1)
    assert list(m.state_dict().keys()) == ['model.0.weight', 'model.1.weight', 'model.1.bias', 'model.1.running_mean',
                                           'model.1.running_var', 'model.1.num_batches_tracked']
    with pytest.raises(Exception):
        m(torch.zeros(1, 1, 4, 4))


@pytest.mark.parametrize("cfg", ["nnyu", "nicvl"])
def test_yaml_surface_and_state_dict_keys(cfg):
    import lsps_amd.trainers as t
    hp = cases.load_hp(cfg)
    tr = t.LSPSTrainer(hp)
    for net, shapes in (('gen', lsps_ref.gen_shapes(hp['gen'])), ('dis', lsps_ref.dis_shapes(hp['dis'])),
                        ('vae', lsps_ref.vae_shapes(hp['vae'])), ('map', lsps_ref.map_shapes(hp['map']))):
        sd = getattr(tr, net).state_dict()
        assert list(sd.keys()) == list(shapes.keys())
        assert all(tuple(sd[k].shape) == tuple(s) for k, s in shapes.items())
    assert sum(p.numel() for p in tr.parameters()) == (71003383 if cfg == 'nnyu' else 71003383 - 60 * (2 * 50 + 1))
    # conv / convT weights re-initialised N(0, 0.02) by gaussian_weights_init, Linear untouched
    w = tr.gen.encode_A[3].model[0].weight
    assert abs(float(w.detach().std()) - 0.02) < 2e-3
    ref_yaml = '/root/reference/exps/%s.yaml' % cfg
    if os.path.exists(ref_yaml):        # the reference's own YAML must load unchanged
        ref_hp = yaml.safe_load(open(ref_yaml))['train']['hyperparameters']
        assert ref_hp == hp
        t.LSPSTrainer(ref_hp)


def test_product_path_fails_loudly_without_gpu():
    if torch.cuda.is_available():
        pytest.skip("has a GPU")
    import lsps_amd.trainers as t
    from lsps_amd import ops
    tr = t.LSPSTrainer(cases.hp_for('tiny'))
    with pytest.raises(Exception):
        tr.cuda(0)
    with pytest.raises(Exception):
        tr.gen(torch.zeros(1, 1, 128, 128), torch.zeros(1, 1, 128, 128))
    with pytest.raises(Exception):
        tr.dis_opt.step()
    with pytest.raises(Exception):
        ops.l1_loss(torch.zeros(4), torch.zeros(4))


def test_product_never_imports_oracle():
    for root, _, files in os.walk(os.path.join(REPO, 'lsps_amd')):
        for f in files:
            if f.endswith('.py'):
                src = open(os.path.join(root, f)).read()
                assert 'import oracle' not in src and 'from oracle' not in src, os.path.join(root, f)


def test_helpers_and_checkpoint_names(tmp_path):
    import lsps_amd.trainers as t
    for name in ('pre_gen_00000010.pkl', 'pre_gen_00000020.pkl', 'pre_dis_00000020.pkl', 'other.txt'):
        (tmp_path / name).write_bytes(b'')
    assert t.get_model_list(str(tmp_path), 'gen').endswith('pre_gen_00000020.pkl')
    assert t.get_model_list(str(tmp_path), 'gen', 0).endswith('pre_gen_00000010.pkl')
    assert t.get_model_list(str(tmp_path / 'missing'), 'gen') is None
    from lsps_amd.trainers.helpers import _compute_true_acc, _compute_fake_acc
    p = torch.tensor([0.1, 0.5, 0.9, 0.7])
    assert _compute_true_acc(p) == 0.75 and _compute_fake_acc(p) == 0.5
