#!/usr/bin/env python
"""Stage-1 driver: trains the pose VAE alone (counterpart of the reference src/pose_train.py:63-191).

Same flags (`--gpu --resume --frac --config --log`), same loop: batches of `batch_size_pose` pose vectors
from both domains (concatenated when `--frac > 0`, pose_train.py:128-129), `vae_sch.step()` every 1000
iterations (:131-132), `vae_update` (:134), loss log every `display` (:137-140), reconstruction read-out
`vae.decode(vae.encode(y)[1])` in mm every `10*image_save_iterations` (:142-178), and `save_vae(prefix, it,
2+frac)` every `4*snapshot_save_iterations` (:181-182).  The cv2 / tensorboardX image dumps are not
reproduced.  Data: seeded synthetic pose vectors (lsps_amd/synth.py) unless loaders are injected.
"""
import argparse
import os
import sys
import time

import numpy as np

from .depth_train import NetConfig, write_loss
from .evaluation import HandposeEvaluation

MAX_ITERATIONS = 200000              # pose_train.py:87 (the YAML's max_iterations is ignored there)


def synthetic_pose_loader(batch_size, label_dim, device, seed, to_tensor):
    """Endless stream of seeded pose batches [batch_size, label_dim] (dataset.pose_only = True, pose_train.py:106-107)."""
    from . import synth
    i = 0
    while True:
        yield to_tensor(synth.make_poses(batch_size, seed + 7919 * i, label_dim), device)
        i += 1


def reconstruction_error(trainer, test_batches):
    """(mean, max) joint error in mm of decode(encode(y).mu) over `test_batches` of
    (labels [n, J*3], com [n, 3], cube [3]) — pose_train.py:150-178."""
    import torch
    gt3d, pr3d = [], []
    with torch.no_grad():
        for labels, com, cube in test_batches:
            pred = trainer.vae.decode(trainer.vae.encode(labels)[1])
            n = labels.shape[0]
            com_np = np.asarray(com.detach().cpu().numpy() if hasattr(com, 'detach') else com, np.float32).reshape(n, 1, 3)
            half = np.asarray(cube, np.float32)[0] / 2.0
            gt3d.append(labels.detach().cpu().numpy().reshape(n, -1, 3) * half + com_np)
            pr3d.append(pred.detach().cpu().numpy().reshape(n, -1, 3) * half + com_np)
    hpe = HandposeEvaluation(np.concatenate(gt3d), np.concatenate(pr3d))
    return hpe.getMeanError(), float(np.nanmax(hpe.err))


def run(opts, trainer_factory=None, loader_a=None, loader_b=None, test_batches=None, device=None, to_tensor=None):
    config = NetConfig(opts.config)
    hp = config.hyperparameters
    batch_size = hp['batch_size_pose']                                                   # :86
    max_iterations = opts.iterations or MAX_ITERATIONS
    frac = opts.frac
    if trainer_factory is None:
        import torch
        from . import trainers
        gpu = int(os.environ.get('LOCAL_RANK', opts.gpu))
        device = torch.device('cuda', gpu)
        trainer = getattr(trainers, hp['trainer'])(hp)                                   # :97-100
        trainer.cuda(gpu)
        to_tensor = lambda a, d: torch.as_tensor(a).to(d)                                # noqa: E731
    else:
        trainer = trainer_factory(hp)
    label_dim = hp['vae']['input_dim']
    seed = config.datasets['train_a']['seed'] if hasattr(config, 'datasets') else 23455
    loader_a = loader_a or synthetic_pose_loader(batch_size, label_dim, device, seed, to_tensor)
    loader_b = loader_b or synthetic_pose_loader(batch_size, label_dim, device, seed + 1, to_tensor)
    os.makedirs(os.path.dirname(config.snapshot_prefix) or '.', exist_ok=True)
    if opts.log:
        os.makedirs(opts.log, exist_ok=True)
    sink = open(os.path.join(opts.log, 'losses_pose.jsonl'), 'a') if opts.log else None
    print('using %.2f percent of the labeled real data' % frac)                          # :121
    import torch
    history, readouts = [], []
    iterations = 0
    start = time.time()
    for labels_a, labels_b in zip(loader_a, loader_b):
        if labels_a.shape[0] != batch_size or labels_b.shape[0] != batch_size:           # :124-125
            continue
        labels = torch.cat((labels_a, labels_b), 0) if frac > 0. else labels_a          # :126-129
        if (iterations + 1) % 1000 == 0:                                                 # :131-132
            trainer.vae_sch.step()
        trainer.vae_update(labels, hp)                                                   # :134
        if (iterations + 1) % config.display == 0:                                       # :137-140
            history.append(write_loss(iterations, max_iterations, trainer, time.time() - start, sink))
            start = time.time()
        if (iterations + 1) % (10 * config.image_save_iterations) == 0 and test_batches:  # :142-178
            mean_err, max_err = reconstruction_error(trainer, test_batches)
            readouts.append((iterations + 1, mean_err, max_err))
            print("Mean error: {}mm, max error: {}mm".format(mean_err, max_err))
        if (iterations + 1) % (4 * config.snapshot_save_iterations) == 0:                # :181-182
            trainer.save_vae(config.snapshot_prefix, iterations, 2 + frac)
        iterations += 1
        if iterations >= max_iterations:
            break
    if sink is not None:
        sink.close()
    return trainer, history, readouts


def build_parser():
    p = argparse.ArgumentParser(description=__doc__.split("\n")[0])
    p.add_argument('--gpu', type=int, default=0, help="gpu id")
    p.add_argument('--resume', type=int, default=0, help="resume training?")
    p.add_argument('--frac', type=float, default=1., help="fraction of real labels to use")
    p.add_argument('--config', type=str, required=True, help="net configuration")
    p.add_argument('--log', type=str, default='', help="log path")
    p.add_argument('--iterations', type=int, default=0, help="stop after this many iterations (default 200000)")
    return p


def main(argv=None):
    run(build_parser().parse_args(argv))


if __name__ == '__main__':
    main(sys.argv[1:])
