"""The driver counterpart on the real HIP trainer: a few pretrain / estimate3 iterations through
lsps_amd.depth_train.run, snapshot save -> resume round trip with the reference's file names, and the
batched evaluation read-out."""
import glob
import os

import numpy as np
import pytest
import torch
import yaml

from lsps_amd import depth_train, synth

pytestmark = pytest.mark.gpu
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _config(tmp_path, **over):
    cfg = yaml.safe_load(open(os.path.join(REPO, 'exps', 'nnyu.yaml')))
    cfg['train']['hyperparameters'] = synth.tiny_hyperparameters(cfg['train']['hyperparameters'], 16, 8)
    cfg['train']['snapshot_prefix'] = str(tmp_path / 'out' / 'pre')
    cfg['train'].update(over)
    p = tmp_path / 'nnyu_small.yaml'
    p.write_text(yaml.safe_dump(cfg))
    return str(p)


def test_pretrain_save_resume_and_estimate(tmp_path):
    if not torch.cuda.is_available():
        pytest.skip("no HIP device")
    cfgp = _config(tmp_path, display=2, snapshot_save_iterations=4, image_save_iterations=3)
    P = depth_train.build_parser().parse_args
    tr, hist = depth_train.run(P(['--config', cfgp, '--mode', 'pretrain', '--batch_size', '4', '--iterations', '4']))
    assert len(hist) == 2 and all(np.isfinite(v) for v in hist[-1].values())
    files = sorted(os.path.basename(f) for f in glob.glob(str(tmp_path / 'out' / '*.pkl')))
    assert files == ['pre_dis_00000004.pkl', 'pre_gen_00000004.pkl']
    sd = torch.load(str(tmp_path / 'out' / 'pre_gen_00000004.pkl'))
    assert len(sd) == 80 and all(v.is_contiguous() for v in sd.values())
    ref_w = tr.gen.state_dict()['encode_A.0.model.0.weight'].clone()

    # resume: weights and iteration counter come back (reference lsps_trainer.py:278-305)
    tr2, _ = depth_train.run(P(['--config', cfgp, '--mode', 'pretrain', '--batch_size', '4', '--iterations', '5',
                                '--resume', '1']))
    assert tr2 is not tr
    # one more iteration happened after loading iteration-4 weights: close to, but not equal to, the saved ones
    w2 = tr2.gen.state_dict()['encode_A.0.model.0.weight']
    assert float((w2 - ref_w).abs().max()) < 5e-4 and float((w2 - ref_w).abs().max()) > 0

    # estimate3 with the batched evaluation read-out
    xb, lb, cb = synth.make_batch(16, 5)
    dev = torch.device('cuda', 0)
    tb = [(torch.as_tensor(xb).to(dev), torch.as_tensor(lb).to(dev), torch.as_tensor(cb).to(dev),
           np.array([300., 300., 300.], np.float32))]
    seen = {}
    from lsps_amd import evaluation

    def ev(trainer, batches, mode_idx, nyu):
        seen['r'] = evaluation.evaluate(trainer, batches, mode_idx, nyu)
        return seen['r']
    tr3, hist3 = depth_train.run(P(['--config', cfgp, '--mode', 'estimate3', '--batch_size', '8', '--iterations', '3']),
                                 test_batches=tb, evaluate_fn=ev)
    assert 'r' in seen and np.isfinite(seen['r'][0]) and 0.0 <= seen['r'][1] <= 100.0
    assert np.isfinite(float(tr3.dis_reg_loss)) and np.isfinite(float(tr3.dis_total_loss))


def test_pose_train_stage1_save_load(tmp_path):
    """Stage 1 on the HIP pose-MLP kernels: the loss falls, `save_vae` writes the reference's file name and
    `load_vae` restores the weights into a fresh trainer (reference src/pose_train.py:122-185, lsps_trainer.py:321-332)."""
    if not torch.cuda.is_available():
        pytest.skip("no HIP device")
    from lsps_amd import pose_train, trainers
    cfgp = _config(tmp_path, display=100, image_save_iterations=30, snapshot_save_iterations=75)
    opts = pose_train.build_parser().parse_args(['--config', cfgp, '--iterations', '300', '--frac', '0.5'])
    dev = torch.device('cuda', 0)
    lb = synth.make_poses(64, 77)
    tb = [(torch.as_tensor(lb).to(dev), np.tile(np.array([[0., 0., 600.]], np.float32), (64, 1)),
           np.array([300., 300., 300.], np.float32))]
    tr, hist, readouts = pose_train.run(opts, test_batches=tb)
    assert len(hist) == 3 and hist[-1]['vae_total_loss'] < hist[0]['vae_total_loss']
    assert [r[0] for r in readouts] == [300] and 0 < readouts[0][1] <= readouts[0][2]
    files = sorted(os.path.basename(f) for f in glob.glob(str(tmp_path / 'out' / '*.pkl')))
    assert files == ['pre_vae_2.50_00000300.pkl']
    hp = yaml.safe_load(open(cfgp))['train']['hyperparameters']
    fresh = trainers.LSPSTrainer(hp)
    fresh.cuda(0)
    fresh.load_vae(str(tmp_path / 'out' / 'pre'), 2.5)
    for k, v in tr.vae.state_dict().items():
        assert torch.equal(v, fresh.vae.state_dict()[k]), k
    # the read-out of the reloaded VAE is the same number
    m2, x2 = pose_train.reconstruction_error(fresh, tb)
    assert abs(m2 - readouts[0][1]) < 1e-4 and abs(x2 - readouts[0][2]) < 1e-3


def test_pretrain_with_gpu_augmentation(tmp_path):
    """`--augment`: every synthetic batch goes through plan_augmentation (host) + lsps_crop_augment (GPU) before the
    trainer sees it; images stay in [-1, 1] with a +1 background, labels follow the augmentation, training runs."""
    if not torch.cuda.is_available():
        pytest.skip("no HIP device")
    cfgp = _config(tmp_path, display=2, snapshot_save_iterations=1000, image_save_iterations=1000)
    P = depth_train.build_parser().parse_args
    dev = torch.device('cuda', 0)
    plain = next(depth_train.synthetic_loader(8, 108, dev, 5, lambda a, d: torch.as_tensor(a).to(d)))
    aug = next(depth_train.synthetic_loader(8, 108, dev, 5, lambda a, d: torch.as_tensor(a).to(d), augment=True))
    assert aug[0].shape == plain[0].shape and float(aug[0].max()) == 1.0 and float(aug[0].min()) >= -1.0
    assert not torch.equal(aug[0], plain[0]) and not torch.equal(aug[1], plain[1])
    tr, hist = depth_train.run(P(['--config', cfgp, '--mode', 'pretrain', '--batch_size', '4', '--iterations', '4', '--augment']))
    assert len(hist) == 2 and all(np.isfinite(v) for v in hist[-1].values())


def test_loss_scalars_materialise_on_read_and_equal_the_synchronous_publication():
    """Round 6: the update methods publish their scalars with ONE asynchronous device -> pinned-host copy and the attributes become
    numpy values when they are READ (no host synchronisation at the end of a step).  Same values as the synchronous path
    (`lazy_scalars = False`), the same reflection the reference's `write_loss` performs (common.py:73-80: `dir()` + `getattr`):
    names appear once an update method has published them and not before; values survive more steps than the ring is deep."""
    import copy
    import cases
    from oracle import lsps_ref
    import lsps_amd.trainers as prod
    from lsps_amd.depth_train import write_loss
    A = cases.NativeAdapter(prod, 'cuda')
    hp = cases.hp_for('tiny')
    sds = cases.make_weights(hp, lsps_ref)
    b = cases.make_inputs(2)
    lat2, lat1 = cases.latent_shape(hp, 4), cases.latent_shape(hp, 2)
    out = []
    for lazy in (True, False):
        tr = A.make_trainer(hp, sds)
        tr.lazy_scalars = lazy
        A.set_train(tr, True)
        assert not [k for k in dir(tr) if 'loss' in k and not k.startswith('_') and not callable(getattr(tr, k))]
        assert not hasattr(tr, 'dis_loss')
        A.dis_update(tr, b, hp, cases.noise(lat2, 1))
        if lazy:
            assert isinstance(tr.__dict__['dis_loss'], tuple)              # in flight: nothing has waited for it
        assert 'dis_loss' in dir(tr) and 'gen_total_loss' not in dir(tr) and not hasattr(tr, 'dis_reg_loss')
        first = float(tr.dis_loss)
        assert isinstance(tr.dis_loss, np.ndarray) and tr.dis_loss.dtype == np.float32 and tr.dis_loss.shape == ()
        A.gen_update(tr, b, hp, (cases.noise(lat2, 2), cases.noise(lat1, 3), cases.noise(lat1, 4)))
        rec = write_loss(0, 1, tr, 0.0)
        assert set(rec) >= {'dis_loss', 'dis_ad_loss', 'dis_true_acc', 'gen_total_loss', 'gen_ll_loss'} and 'dis_reg_loss' not in rec
        keep = copy.deepcopy(rec)
        for it in range(10):                                                # deeper than the ring of pinned buffers (8)
            A.dis_update(tr, b, hp, cases.noise(lat2, 10 + it))
        assert float(tr.gen_total_loss) == keep['gen_total_loss']          # published 10 steps ago by another method, never re-read
        out.append((first, keep, A.scalars(tr)))
    assert out[0][0] == out[1][0] and out[0][1] == out[1][1]
    assert out[0][2].keys() == out[1][2].keys() and all(out[0][2][k] == out[1][2][k] for k in out[0][2])
