"""The reference's one seam for this path is `from trainers import *` (src/depth_train.py:11,
src/pose_train.py:11, src/trainers/__init__.py:5-6).  These tests import the product the way the
reference's driver does — in a FRESH interpreter whose sys.path holds only the entries INTEGRATION.md §1
tells a maintainer to add — and check the namespace, that the module handed out is lsps_amd.trainers
itself, and that nothing of the reference's other packages is shadowed by the recommended recipe."""
import os
import subprocess
import sys
import textwrap

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

BODY = textwrap.dedent('''
    from trainers import *                      # src/depth_train.py:11
    import sys
    # the names the reference's driver uses WITHOUT importing them (depth_train.py:135,145,220)
    for n in ('Variable', 'torch', 'nn', 'os', 'np'):
        assert n in globals(), n
    for n in ('LSPSTrainer', 'SharedResGen', 'SharedResXGen', 'SharedDis', 'poseVAE', 'Mapping', 'LeakyINSResBlock',
              'LeakyReLUConv2d', 'LeakyReLUConvTranspose2d', 'GaussianNoiseLayer', 'gaussian_weights_init',
              'get_model_list'):
        assert n in globals(), n
    import trainers
    import lsps_amd.trainers
    assert trainers is lsps_amd.trainers, (trainers, lsps_amd.trainers)
    assert LSPSTrainer is lsps_amd.trainers.LSPSTrainer
    import trainers.lsps_nets, trainers.common_net, trainers.lsps_trainer, trainers.helpers, trainers.init
    assert trainers.lsps_nets is sys.modules['lsps_amd.trainers.lsps_nets']
    # one module object => one set of process-wide modes and one library handle
    from lsps_amd import ops
    assert trainers.common_net.ops is ops
    import yaml
    ref_yaml = '/root/reference/exps/nnyu.yaml'
    path = ref_yaml if os.path.exists(ref_yaml) else os.path.join(%(repo)r, 'exps', 'nnyu.yaml')
    hp = yaml.safe_load(open(path))['train']['hyperparameters']
    exec("trainer = %%s(hp)" %% hp['trainer'])    # depth_train.py:99 picks the class by name, through exec
    assert type(trainer) is lsps_amd.trainers.LSPSTrainer
    assert sum(p.numel() for p in trainer.parameters()) == 71003383
    assert len(trainer.gen.state_dict()) == 80 and len(trainer.dis.state_dict()) == 20
    print("DROPIN-OK", path)
''') % {'repo': REPO}


def _run(path_entries, extra=''):
    # -S -E -s: no user site, no PYTHON* env; cwd outside the repo so '' on sys.path finds nothing of ours
    prog = "import sys\n" + "".join("sys.path.insert(0, %r)\n" % p for p in path_entries) + BODY + extra
    env = {k: v for k, v in os.environ.items() if k != 'PYTHONPATH'}
    r = subprocess.run([sys.executable, '-c', prog], cwd='/tmp', env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    assert 'DROPIN-OK' in r.stdout
    return r.stdout


def test_documented_recipe_two_path_entries():
    """INTEGRATION.md §1, variant B: repo root + package dir, exactly as r3's recipe read."""
    _run([REPO, os.path.join(REPO, 'lsps_amd')])


def test_package_dir_alone_bootstraps_the_repo_root():
    _run([os.path.join(REPO, 'lsps_amd')])


def test_dropin_shim_shadows_only_trainers():
    """INTEGRATION.md §1, variant A (recommended): <repo>/dropin in front of sys.path.  The reference's driver also does
    `from data import *`, `from utils import *`, `from common import ...` (depth_train.py:8-12): those names must NOT resolve
    into this repo."""
    extra = textwrap.dedent('''
        import importlib.util
        for name in ('data', 'utils', 'common', 'net_config', 'dist', 'ops', 'optim'):
            spec = importlib.util.find_spec(name)
            assert spec is None or %(repo)r not in (spec.origin or ''), (name, spec.origin)
        print("NO-SHADOW-OK")
    ''') % {'repo': REPO}
    out = _run([os.path.join(REPO, 'dropin')], extra)
    assert 'NO-SHADOW-OK' in out


@pytest.mark.skipif(not os.path.isdir('/root/reference/src'), reason="reference tree not on this box")
def test_dropin_in_front_of_the_reference_src_dir():
    """With the reference's own src/ on sys.path (how depth_train.py is run: cwd = src/), the shim in front of it wins for
    `trainers` and src/ keeps `data`, `utils`."""
    extra = textwrap.dedent('''
        import importlib.util
        assert importlib.util.find_spec('data').origin.startswith('/root/reference/src/data')
        assert importlib.util.find_spec('utils').origin.startswith('/root/reference/src/utils')
        assert trainers.__file__.startswith(%(repo)r)
        print("ORDER-OK")
    ''') % {'repo': REPO}
    out = _run(['/root/reference/src', os.path.join(REPO, 'dropin')], extra)
    assert 'ORDER-OK' in out


def test_loss_scalars_reflect_like_the_references_attributes():
    """Round 6: `dis_loss` & co. are class-level properties (published asynchronously on the GPU, materialised on read).  For a caller
    they must behave like the reference's plain attributes (lsps_trainer.py:73,132-140,198-199,214-217,260-261; common.py:73-80 reflects
    with dir() + getattr): absent until an update method published them, numpy values afterwards, assignable, deletable, deep-copyable."""
    import copy
    import sys
    import numpy as np
    sys.path.insert(0, os.path.join(REPO, 'tests', 'golden'))
    import cases
    import lsps_amd.trainers as prod
    from lsps_amd.depth_train import write_loss
    tr = prod.LSPSTrainer(cases.hp_for('tiny'))
    assert not hasattr(tr, 'dis_loss') and 'dis_loss' not in dir(tr) and 'dis_loss' not in vars(tr)
    assert write_loss(0, 1, tr, 0.0).keys() == {'iteration', 'sec_per_display'}
    tr.dis_loss = np.asarray(1.5, dtype=np.float32)                    # what _finish_step's synchronous path does
    tr.gen_total_loss = np.asarray(2.5, dtype=np.float32)
    assert 'dis_loss' in dir(tr) and 'dis_loss' in vars(tr) and float(tr.dis_loss) == 1.5
    rec = write_loss(0, 1, tr, 0.0)
    assert rec['dis_loss'] == 1.5 and rec['gen_total_loss'] == 2.5 and 'dis_reg_loss' not in rec
    clone = copy.deepcopy(tr)
    assert float(clone.dis_loss) == 1.5 and not hasattr(clone, 'vae_total_loss')
    del tr.dis_loss
    assert not hasattr(tr, 'dis_loss') and float(clone.dis_loss) == 1.5
    # no other non-callable public attribute may contain 'loss' / 'acc' (SURVEY 8(b)): the reflective logger would print it
    extra = [a for a in dir(tr) if ('loss' in a or 'acc' in a) and not a.startswith('_') and not callable(getattr(tr, a))]
    assert extra == ['gen_total_loss'], extra
