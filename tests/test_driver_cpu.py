"""BASELINE config 1 (plumbing): the driver loop (lsps_amd/depth_train.py, counterpart of the reference
src/depth_train.py:140-265) for 10 pretrain iterations at bs=8 on synthetic 128x128 depth, CPU, with the
oracle injected as the trainer (no GPU here; the HIP trainer takes the same path on the GPU box)."""
import os

import numpy as np
import torch
import yaml

import cases
from lsps_amd import depth_train, evaluation, synth
from oracle import lsps_ref

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


class _CountingSched(object):
    def __init__(self):
        self.n = 0

    def step(self):
        self.n += 1


class _OracleTrainer(lsps_ref.RefTrainer):
    saved = []

    def __init__(self, hp):
        super(_OracleTrainer, self).__init__(hp, literal=False)
        sd = cases.make_weights(hp, lsps_ref)
        for net in ('gen', 'dis', 'vae', 'map'):
            getattr(self, net).load_state_dict(sd[net])
        self.dis_sch, self.gen_sch = _CountingSched(), _CountingSched()

    def save(self, prefix, iterations):
        _OracleTrainer.saved.append('%s_gen_%08d.pkl' % (prefix, iterations + 1))


def _config(tmp_path, **over):
    cfg = yaml.safe_load(open(os.path.join(REPO, 'exps', 'nnyu.yaml')))
    cfg['train']['hyperparameters'] = synth.tiny_hyperparameters(cfg['train']['hyperparameters'])
    cfg['train']['snapshot_prefix'] = str(tmp_path / 'out' / 'pre')
    cfg['train'].update(over)
    p = tmp_path / 'nnyu_tiny.yaml'
    p.write_text(yaml.safe_dump(cfg))
    return str(p)


def _t(a, d):
    return torch.as_tensor(a)


def test_pretrain_10_iterations_bs8_cpu(tmp_path):
    torch.set_num_threads(8)
    _OracleTrainer.saved = []
    cfgp = _config(tmp_path, display=5, snapshot_save_iterations=10)
    opts = depth_train.build_parser().parse_args(['--config', cfgp, '--mode', 'pretrain', '--batch_size', '8',
                                                  '--iterations', '10', '--log', str(tmp_path / 'log')])
    tr, hist = depth_train.run(opts, trainer_factory=_OracleTrainer, device='cpu', to_tensor=_t)
    assert len(hist) == 2 and hist[-1]['iteration'] == 10
    for k in ('dis_loss', 'dis_ad_loss', 'dis_feat_loss', 'dis_true_acc', 'dis_fake_acc', 'gen_total_loss',
              'gen_enc_loss', 'gen_ll_loss', 'gen_ad_loss'):
        assert np.isfinite(hist[-1][k]), k
    assert hist[-1]['gen_enc_loss'] < hist[0]['gen_enc_loss']              # it trains: the latent KL term falls
    assert _OracleTrainer.saved == [str(tmp_path / 'out' / 'pre') + '_gen_00000010.pkl']
    assert tr.dis_sch.n == 0                                               # pretrain: schedulers only every 1000 its
    assert os.path.exists(tmp_path / 'log' / 'losses.jsonl')


def test_estimate3_cadence_and_eval(tmp_path):
    torch.set_num_threads(8)
    cfgp = _config(tmp_path, display=100, image_save_iterations=100, snapshot_save_iterations=1000)
    opts = depth_train.build_parser().parse_args(['--config', cfgp, '--mode', 'estimate3', '--batch_size', '8',
                                                  '--iterations', '100'])
    xb, lb, cb = synth.make_batch(16, 5)
    test_batches = [(torch.as_tensor(xb), torch.as_tensor(lb), cb, np.array([300., 300., 300.], np.float32))]
    seen = {}

    def eval_fn(trainer, batches, mode_idx, nyu):
        class _T(object):       # adapt the oracle nets to the evaluate() protocol
            pass
        t = _T()
        t.dis, t.vae = trainer.dis, trainer.vae
        t.dis.eval = lambda: None
        seen['res'] = evaluation.evaluate(t, batches, mode_idx, nyu)
        return seen['res']

    tr, hist = depth_train.run(opts, trainer_factory=_OracleTrainer, device='cpu', to_tensor=_t,
                               test_batches=test_batches, evaluate_fn=eval_fn)
    assert tr.dis_sch.n == 1 and tr.gen_sch.n == 0                          # estimate: dis_sch every 100 its
    mean_err, pct = seen['res']
    ref = lsps_ref.joint_readout(tr.dis, tr.vae, torch.as_tensor(xb), torch.as_tensor(lb), cb,
                                 np.array([300., 300., 300.], np.float32))
    assert abs(mean_err - ref['mean_err']) < 1e-3 and abs(pct - 100.0 * ref['frames_within_40'] / 16) < 1e-9


def test_evaluation_formulas():
    gt = np.zeros((3, 4, 3))
    pr = np.zeros((3, 4, 3))
    pr[0, 1] = [3, 4, 0]        # 5 mm
    pr[1, 2] = [0, 0, 50]       # 50 mm
    e = evaluation.HandposeEvaluation(gt, pr)
    assert abs(e.getMeanError() - (5 / 4 + 50 / 4) / 3) < 1e-12
    assert e.getNumFramesWithinMaxDist(40) == 2
    assert list(e.getWorstJoint()[:2]) == [1, 2]
    assert np.allclose(e.getMaxErrorOverSeq(), [5, 50, 0])


def test_pose_train_cadence_cpu(tmp_path):
    """Stage 1 (reference src/pose_train.py:122-185): both domains' poses concatenated when frac > 0, vae_sch
    every 1000 its, reconstruction read-out every 10*image_save_iterations, save_vae name with 2+frac."""
    from lsps_amd import pose_train
    torch.set_num_threads(8)
    saved = []

    class _T(_OracleTrainer):
        def __init__(self, hp):
            super(_T, self).__init__(hp)
            self.vae_sch = _CountingSched()
            self.seen = []

        def vae_update(self, y, hp):
            self.seen.append(tuple(y.shape))
            return super(_T, self).vae_update(y, hp)

        def save_vae(self, prefix, iterations, frac):
            saved.append('%s_vae_%.2f_%08d.pkl' % (prefix, frac, iterations + 1))

    cfgp = _config(tmp_path, display=500, image_save_iterations=100, snapshot_save_iterations=250)
    opts = pose_train.build_parser().parse_args(['--config', cfgp, '--iterations', '1000', '--frac', '1.0',
                                                 '--log', str(tmp_path / 'log')])
    lb = synth.make_poses(32, 77)
    tb = [(torch.as_tensor(lb), np.tile(np.array([[0., 0., 600.]], np.float32), (32, 1)),
           np.array([300., 300., 300.], np.float32))]
    tr, hist, readouts = pose_train.run(opts, trainer_factory=_T, device='cpu', to_tensor=_t, test_batches=tb)
    bs = yaml.safe_load(open(cfgp))['train']['hyperparameters']['batch_size_pose']
    assert set(tr.seen) == {(2 * bs, 108)} and len(tr.seen) == 1000
    assert tr.vae_sch.n == 1
    assert [r[0] for r in readouts] == [1000]
    assert saved == [str(tmp_path / 'out' / 'pre') + '_vae_3.00_00001000.pkl']
    assert len(hist) == 2 and hist[-1]['vae_total_loss'] < hist[0]['vae_total_loss']
    mean_err, max_err = readouts[0][1], readouts[0][2]
    assert 0 < mean_err <= max_err < 300.0          # mm, poses are within the 300 mm cube
