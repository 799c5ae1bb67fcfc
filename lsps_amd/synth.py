"""Synthetic "NYU-shape" batches and seeded weights.

There is no network / dataset here, so the bench, the smoke test and the parity tests all
run on synthetic 128x128 depth crops shaped like the reference's data contract
(``src/data/dataset_hand2.py:27-31,111-116``: float32 [N,1,128,128] in [-1,1], background
exactly +1; labels float32 [N,108] = joints/(cube/2); com [N,3]).  Everything is generated
with ``numpy.random.RandomState`` so it is bit-identical on every box.
"""
import numpy as np

YAML_SEED = 23455          # exps/nnyu.yaml:63 (datasets.train_a.seed)


def make_batch(n, seed=YAML_SEED, label_dim=108, size=128):
    """Returns (images[n,1,S,S], labels[n,label_dim], com[n,3]) float32 numpy arrays."""
    rs = np.random.RandomState(seed)
    yy, xx = np.mgrid[0:size, 0:size].astype(np.float32)
    imgs = np.ones((n, 1, size, size), np.float32)
    for i in range(n):
        cy, cx = size / 2 + rs.uniform(-6, 6), size / 2 + rs.uniform(-6, 6)
        ry, rx = rs.uniform(0.26, 0.34) * size, rs.uniform(0.26, 0.34) * size
        mask = ((yy - cy) / ry) ** 2 + ((xx - cx) / rx) ** 2 < 1.0
        d = rs.uniform(-0.9, 0.6, size=(size // 8, size // 8)).astype(np.float32)
        d = np.kron(d, np.ones((8, 8), np.float32))
        # cheap separable smoothing (box, twice)
        for _ in range(2):
            d = (d + np.roll(d, 1, 0) + np.roll(d, -1, 0) + np.roll(d, 2, 0) + np.roll(d, -2, 0)) / 5.0
            d = (d + np.roll(d, 1, 1) + np.roll(d, -1, 1) + np.roll(d, 2, 1) + np.roll(d, -2, 1)) / 5.0
        imgs[i, 0][mask] = np.clip(d, -1.0, 1.0)[mask]
    labels = np.clip(rs.normal(0.0, 0.3, size=(n, label_dim)), -1.0, 1.0).astype(np.float32)
    com = np.tile(np.array([[0.0, 0.0, 600.0]], np.float32), (n, 1))
    return imgs, labels, com


def make_poses(n, seed=YAML_SEED, label_dim=108):
    """Pose vectors only, [n, label_dim] float32 ~ N(0, 0.3) clipped to [-1, 1] (the stage-1 loaders run with
    ``pose_only = True``, ``src/pose_train.py:106-107``)."""
    rs = np.random.RandomState(seed)
    return np.clip(rs.normal(0.0, 0.3, size=(n, label_dim)), -1.0, 1.0).astype(np.float32)


def make_state_dict(shapes, seed):
    """Seeded weights for a ``key -> shape`` table, following the reference's init rules:
    conv / conv-transpose weights ~ N(0, 0.02) (``src/trainers/init.py:8-12``), biases and
    Linear weights ~ U(+-1/sqrt(fan_in)) (torch defaults), poseVAE mu/sigma heads ~ N(0, 0.002)
    (``src/trainers/lsps_nets.py:55-59``)."""
    rs = np.random.RandomState(seed)
    out = {}
    last_fan_in = 1
    for k, shp in shapes.items():
        shp = tuple(shp)
        if k.startswith('en_mu') or k.startswith('en_sigma'):
            v = rs.normal(0.0, 0.002, size=shp)
        elif k.endswith('.weight') and len(shp) == 4:
            v = rs.normal(0.0, 0.02, size=shp)
            last_fan_in = shp[1] * shp[2] * shp[3]
        elif k.endswith('.weight'):
            last_fan_in = shp[1]
            b = 1.0 / np.sqrt(last_fan_in)
            v = rs.uniform(-b, b, size=shp)
        else:
            b = 1.0 / np.sqrt(last_fan_in)
            v = rs.uniform(-b, b, size=shp)
        out[k] = v.astype(np.float32)
    return out


def tiny_hyperparameters(hp, gen_ch=8, dis_ch=4):
    """Same topology as the YAML config with narrow channels (for CPU-sized parity cases)."""
    import copy
    hp = copy.deepcopy(hp)
    hp['gen']['ch'] = gen_ch
    hp['dis']['ch'] = dis_ch
    hp['map']['output_ch'] = gen_ch * 2 ** (hp['gen']['n_enc_front_blk'] - 1)   # Mapping emits the latent's channels
    return hp
