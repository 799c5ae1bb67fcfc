#!/usr/bin/env python
"""Python-3 counterpart of the reference driver (src/depth_train.py:63-265) for the HIP trainer.

Same flags (`--gpu --resume --frac --idx --config --mode --log`), same loop order and scheduler
cadence, same snapshot names and the same periodic evaluation read-out; the parts of the reference
driver that need the NYU/ICVL datasets, cv2 or tensorboardX (image dumps, AVI, HTML index) are not
reproduced.  Data: `--data synthetic` (seeded NYU-shape batches, lsps_amd/synth.py) — a real dataset
plugs in as any iterable of `(images, labels, com)` pairs through `run(..., loader_a=, loader_b=)`.
Additions: `--batch_size` (the reference hard-wires 1 in pretrain mode, depth_train.py:85),
`--iterations`, and one-process-per-GPU data parallelism when launched by torchrun.
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import yaml


class NetConfig(object):
    """YAML `train:` section -> attributes (reference: src/utils/net_config.py:9-20, without exec)."""

    def __init__(self, path):
        with open(path, 'r') as f:
            for doc in yaml.safe_load_all(f):
                for k, v in (doc or {}).get('train', {}).items():
                    setattr(self, k, v)


def synthetic_loader(batch_size, label_dim, device, seed, to_tensor, augment=False):
    """Endless stream of seeded NYU-shape batches (stands in for get_data_loader, common.py:16-17).
    `augment`: run the reference's per-sample augmentation (dataset_hand2.py:331-364, modes none / com / rot) on every
    batch — geometry planned on the host in the dataset's RNG order, pixels warped on the GPU in one launch
    (lsps_amd/data.py)."""
    from . import synth
    pipe = cam = rng = None
    if augment:
        from . import data as ldata
        pipe = ldata.CropPipeline(device)
        cam = ldata.NYU_CAMERA if label_dim == 108 else ldata.ICVL_CAMERA
        cube = np.full((3,), 300.0 if label_dim == 108 else 250.0, np.float32)
        rng = np.random.RandomState(seed)                                                # dataset_hand2.py:264
    i = 0
    while True:
        x, l, c = synth.make_batch(batch_size, seed + 7919 * i, label_dim)
        if augment:
            plans = []
            for k in range(batch_size):
                com2d = cam.to_img(c[k])
                M = np.asarray(ldata.crop_transform(cam, com2d, cube, (x.shape[3], x.shape[2])), 'float32')
                plans.append(ldata.plan_augmentation(cam, l[k].reshape(-1, 3) * (cube[2] / 2.), com2d, cube, M,
                                                     list(ldata.DEFAULT_AUG_MODES), rng, (x.shape[3], x.shape[2])))
            xt = pipe.augment(to_tensor(x, device), plans)
            l = np.stack([p.label.reshape(-1) for p in plans])
            c = np.stack([p.com3D for p in plans])
            yield xt, to_tensor(l, device), to_tensor(c, device)
        else:
            yield to_tensor(x, device), to_tensor(l, device), to_tensor(c, device)
        i += 1


def write_loss(iterations, max_iterations, trainer, elapsed, sink=None):
    """Reflects over the trainer's `*loss*` / `*acc*` scalars like common.py:71-80 (stdout / JSONL)."""
    members = [a for a in dir(trainer) if not callable(getattr(trainer, a)) and not a.startswith('__')
               and ('loss' in a or 'acc' in a) and not a.startswith('_')]
    rec = {'iteration': iterations + 1, 'sec_per_display': elapsed}
    for m in members:
        try:
            rec[m] = float(np.asarray(getattr(trainer, m)).reshape(-1)[0])
        except (TypeError, ValueError):
            continue
    print("Iteration: %08d/%08d  %.3f s  " % (iterations + 1, max_iterations, elapsed) +
          " ".join("%s=%.5g" % (k, v) for k, v in rec.items() if k not in ('iteration', 'sec_per_display')))
    if sink is not None:
        sink.write(json.dumps(rec) + "\n")
        sink.flush()
    return rec


def run(opts, trainer_factory=None, loader_a=None, loader_b=None, test_batches=None, device=None, to_tensor=None,
        evaluate_fn=None):
    """The hot loop of depth_train.py:140-265.  `trainer_factory(hyperparameters)` defaults to the HIP
    LSPSTrainer; tests inject the CPU oracle here to check the plumbing without a GPU."""
    config = NetConfig(opts.config)
    hp = config.hyperparameters
    mode_idx = int(opts.mode[-1]) if 'estimate' in opts.mode else None
    batch_size = opts.batch_size or (hp['batch_size'] if 'estimate' in opts.mode else 1)     # :85
    max_iterations = opts.iterations or hp['max_iterations']
    if trainer_factory is None:
        import torch
        from . import trainers
        from . import dist as ldist
        gpu = int(os.environ.get('LOCAL_RANK', opts.gpu))
        device = torch.device('cuda', gpu)
        # Bind this rank to ITS GPU before the first collective: resume() below (reference order, depth_train.py:103-107,
        # is resume THEN cuda) already broadcasts rank 0's iteration count, and with every rank still on cuda:0 RCCL would
        # see duplicate devices.
        torch.cuda.set_device(device)
        if int(os.environ.get('WORLD_SIZE', '1')) > 1 and not torch.distributed.is_initialized():
            os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
            # torch's flight recorder is OFF by default (profiles/r4b_drain_probe.txt); dist.drain_watchdog needs it to
            # confirm that RCCL's watchdog holds no eager work before a data-parallel step is captured into a hipGraph
            os.environ.setdefault('TORCH_FR_BUFFER_SIZE', os.environ.get('TORCH_NCCL_TRACE_BUFFER_SIZE', '2000'))   # (torch < 2.9: TORCH_NCCL_TRACE_BUFFER_SIZE)
            os.environ.setdefault('TORCH_NCCL_TRACE_BUFFER_SIZE', os.environ['TORCH_FR_BUFFER_SIZE'])
            torch.distributed.init_process_group('nccl', device_id=device)
        trainer = getattr(trainers, hp['trainer'])(hp)                                   # :99-102
        iterations = 0
        if opts.resume == 1:                                                             # :109-113
            iterations = trainer.resume(config.snapshot_prefix, idx=-1, load_opt=True)
            for _ in range(iterations // 1000):
                trainer.dis_sch.step()
                trainer.gen_sch.step()
        trainer.cuda(gpu)
        if getattr(opts, 'dtype', 'f32') != 'f32':
            from . import ops as _ops
            _ops.set_math_mode(opts.dtype)  # process-wide: bf16 MFMA conv path on bf16 activations
        if getattr(opts, 'graphs', False):
            trainer.use_graphs(True)        # hipGraph replay of the update steps (single process)
        try:                                                                             # :116-124
            trainer.load_vae(config.snapshot_prefix, 2 + opts.frac if mode_idx in (3, 4) else opts.frac)
        except Exception:
            print('Failed to load the parameters of vae')
        if 'estimate' in opts.mode and opts.idx != 0:                                    # :126-128
            trainer.resume(config.snapshot_prefix, idx=opts.idx, est=(mode_idx == 5))
        to_tensor = lambda a, d: torch.as_tensor(a).to(d)                                # noqa: E731
        seed_shift = 100 * ldist.rank()
        if evaluate_fn is None:
            from .evaluation import evaluate as evaluate_fn
    else:
        trainer = trainer_factory(hp)
        iterations = 0
        seed_shift = 0
    label_dim = hp['vae']['input_dim']
    seed = config.datasets['train_a']['seed'] if hasattr(config, 'datasets') else 23455
    aug = bool(getattr(opts, 'augment', False)) and trainer_factory is None
    loader_a = loader_a or synthetic_loader(batch_size, label_dim, device, seed + seed_shift, to_tensor, aug)
    loader_b = loader_b or synthetic_loader(batch_size, label_dim, device, seed + 1 + seed_shift, to_tensor, aug)
    os.makedirs(os.path.dirname(config.snapshot_prefix) or '.', exist_ok=True)
    if opts.log:
        os.makedirs(opts.log, exist_ok=True)
    is_rank0 = int(os.environ.get('RANK', '0')) == 0          # replicas log identical (rank-averaged) scalars: one writer
    sink = open(os.path.join(opts.log, 'losses.jsonl'), 'a') if (opts.log and is_rank0) else None
    history = []
    best_err, best_acc = 100.0, 0.0
    start = time.time()
    for (images_a, labels_a, com_a), (images_b, labels_b, com_b) in zip(loader_a, loader_b):
        if images_a.shape[0] != batch_size or images_b.shape[0] != batch_size:           # :143-144
            continue
        if hasattr(trainer.dis, 'train'):
            trainer.dis.train()                                                          # :152
        if opts.mode == 'pretrain':
            if (iterations + 1) % 1000 == 0:                                             # :154-157
                trainer.dis_sch.step()
                trainer.gen_sch.step()
            trainer.dis_update(images_a, labels_a, images_b, labels_b, com_a, com_b, hp)
            image_outputs = trainer.gen_update(images_a, labels_a, images_b, labels_b, hp)
        else:
            if (iterations + 1) % 100 == 0:                                              # :163-164
                trainer.dis_sch.step()
            image_outputs = trainer.post_update(images_a, labels_a, images_b, labels_b, com_a, com_b, mode_idx, hp)
        if hasattr(trainer, 'assemble_outputs'):
            trainer.assemble_outputs(images_a, images_b, image_outputs)                   # :161,166
        if (iterations + 1) % config.display == 0 and is_rank0:                          # :169-172
            history.append(write_loss(iterations, max_iterations, trainer, time.time() - start, sink))
            start = time.time()
        if (iterations + 1) % config.image_save_iterations == 0 and 'estimate' in opts.mode and test_batches \
                and evaluate_fn is not None:                                              # :185-253
            mean_err, over_40 = evaluate_fn(trainer, test_batches, mode_idx, 'nyu' in opts.config)   # depth_train.py:231
            best_err, best_acc = min(best_err, mean_err), max(best_acc, over_40)
            print("------------ Mean err: {:.4f} ({:.4f}) mm, Max over 40mm: {:.2f} ({:.2f}) %".format(
                mean_err, best_err, over_40, best_acc))
        if (iterations + 1) % config.snapshot_save_iterations == 0:                      # :257-261
            trainer.save(config.snapshot_prefix + ('_est' if 'estimate' in opts.mode else ''), iterations)
        iterations += 1
        if iterations >= max_iterations:
            break
    if sink is not None:
        sink.close()
    return trainer, history


def build_parser():
    p = argparse.ArgumentParser(description=__doc__.split("\n")[0])
    p.add_argument('--gpu', type=int, default=0, help="gpu id")
    p.add_argument('--resume', type=int, default=0, help="resume training?")
    p.add_argument('--frac', type=float, default=1., help="fraction of real labels to use")
    p.add_argument('--idx', type=int, default=-1, help="idx pretrain")
    p.add_argument('--config', type=str, required=True, help="net configuration")
    p.add_argument('--mode', type=str, required=True, help="pretrain / estimate<k>")
    p.add_argument('--log', type=str, default='', help="log path")
    p.add_argument('--batch_size', type=int, default=0, help="override (reference: 1 in pretrain, YAML in estimate)")
    p.add_argument('--iterations', type=int, default=0, help="stop after this many iterations (default: YAML max_iterations)")
    p.add_argument('--data', type=str, default='synthetic', choices=['synthetic'])
    p.add_argument('--dtype', type=str, default='f32', choices=['f32', 'bf16'],
                   help="'bf16': bf16 MFMA conv path on bf16 activations (BASELINE config 5, e.g. with exps/nicvl.yaml); default exact f32")
    p.add_argument('--graphs', action='store_true', help="replay dis_update / gen_update / post_update from hipGraphs")
    p.add_argument('--augment', action='store_true', help="augment every batch as the datasets do (geometry on the host, pixels on the GPU)")
    return p


def main(argv=None):
    opts = build_parser().parse_args(argv)
    run(opts)


if __name__ == '__main__':
    main(sys.argv[1:])
