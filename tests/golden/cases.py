"""Implementation-agnostic parity cases for the depth path.

The same case list is executed by three adapters:
  * ``RefAdapter``    (tests/golden/make_golden.py) — the REAL reference, imported from
                       /root/reference in the build container; its outputs are committed as
                       ``tests/golden/golden_*.npz``;
  * ``NativeAdapter`` over ``oracle.lsps_ref``  — the CPU restatement (tests, CPU);
  * ``NativeAdapter`` over ``lsps_amd.trainers`` — the HIP product path (tests, ``-m gpu``).

Inputs, weights and noise are never stored: they are regenerated from numpy seeds
(``lsps_amd.synth``), which is bit-stable across machines.  Only outputs are stored, as
full tensors when small and as digests (mean / abs-max / L2 / 256 seeded samples) when large.
"""
import copy
import os
from collections import OrderedDict

import numpy as np
import yaml

from lsps_amd import synth

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
FULL_STORE_LIMIT = 4096          # tensors up to this many elements are stored in full
PARAM_ABS = 6e-4                 # 2 steps x (2*lr = 2e-4 for dis/gen; vae lr*10 handled by its own scale) + margin


def load_hp(name='nnyu'):
    with open(os.path.join(REPO, 'exps', name + '.yaml')) as f:
        return yaml.safe_load(f)['train']['hyperparameters']


def hp_for(config):
    hp = load_hp('nnyu')
    if config == 'tiny':
        hp = synth.tiny_hyperparameters(hp)
    return hp


def noise(shape, seed, std=1.0):
    return (np.random.RandomState(seed).standard_normal(size=shape) * std).astype(np.float32)


def digest(x):
    x = np.asarray(x)
    flat = x.astype(np.float64).ravel()
    d = OrderedDict()
    d['shape'] = np.asarray(x.shape, np.int64)
    if flat.size <= FULL_STORE_LIMIT:
        d['full'] = x.astype(np.float32)
        return d
    idx = np.random.RandomState(12345).randint(0, flat.size, size=256)
    d['mean'] = np.float64(flat.mean())
    d['absmax'] = np.float64(np.abs(flat).max())
    d['l2'] = np.float64(np.sqrt((flat ** 2).sum()))
    d['sample'] = flat[idx].astype(np.float32)
    return d


def flatten(results):
    """{case: {name: array}} -> flat npz dict of digests."""
    out = {}
    for case, vals in results.items():
        for name, v in vals.items():
            for k, a in digest(v).items():
                out['%s/%s/%s' % (case, name, k)] = a
    return out


def dead_bias_keys(golden):
    """Biases of convs that feed an affine-free InstanceNorm (both convs of every LeakyINSResBlock,
    common_net.py:160-181) are mathematically cancelled: their gradient is exactly 0 and what the
    reference computes for it is pure round-off (~1e-8), which Adam then amplifies to +-lr steps.
    Nothing observable depends on them; their GRADIENTS are excluded from comparisons, their post-step VALUES are compared
    with the rule in `compare` (the weight-decay term decides the sign of most elements)."""
    names = set(k.split('/')[1] for k in golden if k.count('/') >= 2)
    dead = set()
    for nme in names:
        if nme.endswith('.model.3.bias'):
            dead.add(nme)
            dead.add(nme[:-len('.model.3.bias')] + '.model.0.bias')
            dead.add(nme[:-len('.model.3.bias')] + '.model.6.bias')      # ResNeXt block: third conv, also under IN
    return dead


# gradients, robust criteria (relative; see `compare`): L2 digest, mean, 1 - cosine, and two quantiles of the element error.
# What the bounds are made of (profiles/r4e_gradient_criterion_*.txt):
#   * L2 / mean / cosine catch a SYSTEMATIC error (a mis-scaled or mis-directed gradient) and sit at 1e-3; measured worst over
#     every golden step case and every conv algorithm of the product: L2 5.4e-4, 1 - cosine 2.5e-5 (restatement of the
#     reference on identical torch kernels: 1.8e-4, 1.1e-6);
#   * the quantiles bound the ISOLATED flips (LeakyReLU kinks, sign() of the L1 losses: a forward difference of 1e-6 flips a
#     different set of them).  They are noise, not accuracy: the restatement itself has q90 5.2e-4 / q99 1.65e-3, the product
#     between q90 8.9e-4 / q99 1.8e-3 (direct and F(2x2,3x3) kernels: what a differentiated pass at these batch sizes runs)
#     and q90 1.1e-3 / q99 3.2e-3 with the F(4x4,3x3) kernels forced (forward error 8e-6 instead of 5e-7; a reduction-split
#     F(4x4,3x3) forward at N = 2 measured q90 1.4e-3 / q99 6.6e-3 and is therefore kept to passes without a backward:
#     include/lsps_hip.h, lsps_conv2d_in_fwd_nograd); hence q90 <= 2e-3, q99 <= 5e-3.
GRAD_ROBUST = {'l2': 1e-3, 'mean': 1e-3, 'cosine': 1e-3, 'q99': 5e-3, 'q90': 2e-3}
# ResNeXt generator at 8-64 channels: three LeakyReLU / InstanceNorm stages per block, restatement-vs-reference quantiles 1.4e-2 (99 %) and 1.25e-2 (90 %)
GRAD_ROBUST_RESX = {'l2': 2e-3, 'mean': 1e-3, 'cosine': 1e-3, 'q99': 3e-2, 'q90': 2e-2}


def compare(results, golden, rtol, atol_scale=1.0, skip=(), grad_rtol=None, grad_robust=GRAD_ROBUST, report=None):
    """Returns list of (key, err, tol) failures. Error is relative to the tensor's abs-max.
    ``grad_rtol`` (default = rtol) applies to the MAX element error of '*.grads' cases: gradients of this model are
    discontinuous in the activations (sign() of the L1 losses, LeakyReLU kinks), so fp32
    round-off differences between two correct implementations flip isolated elements and show
    up at the 1e-3..1e-2 level of a gradient tensor's abs-max (measured: reference vs. its own
    restatement on the same torch CPU kernels reaches 3.6e-3 at ch=64).
    Isolated flips cannot hide a systematically wrong gradient, though, so every gradient tensor must
    ALSO pass the robust criteria of ``grad_robust`` (GRAD_ROBUST): its L2 norm within 1e-3 relative, its mean within 1e-3
    of abs-max, the cosine between its stored elements (full tensor or the 256 seeded samples) and the reference's within
    1e-3 of 1, 90 % of those elements within 2e-3 and 99 % within 5e-3 of abs-max (isolated kink flips; the restatement of
    the reference on identical kernels already has a 99th percentile of 1.65e-3).  ``report`` (a dict) receives the worst value per criterion."""
    bad = []
    worst = 0.0
    rep = report if report is not None else {}
    flat = flatten(results)
    dead = dead_bias_keys(golden)
    dead_off = {}
    for key, g in golden.items():
        if any(key.startswith(s) for s in skip):
            continue
        parts = key.split('/')
        if parts[1] in dead and 'grads' in parts[0]:
            continue                       # the reference's gradient there is round-off noise around an exact 0
        if parts[1] in dead and 'params' in parts[0]:
            # ... but its Adam step is not noise: g = noise + weight_decay * p is dominated by the decay term unless |p| is
            # tiny, so the bias moves by lr * sign(p) per step.  The product feeds Adam an exact zero gradient + the decay:
            # most elements land on the reference's value, the sign-ambiguous rest within 2 lr per step.
            # A single tensor can be mostly sign-ambiguous (the reference vs. its own restatement: 56 % of one bias), so
            # the fraction is judged over all dead biases of a case (restatement: 1 - 4 %; biases that never move: 100 %).
            if key in flat and key.rsplit('/', 1)[1] in ('full', 'sample') and g.size:
                d = np.abs(np.asarray(flat[key], np.float64) - np.asarray(g, np.float64))
                if float(d.max()) > PARAM_ABS:
                    bad.append((key, float(d.max()), PARAM_ABS))
                dead_off.setdefault(parts[0], []).append(float((d > 5e-6).mean()))
            continue
        if key not in flat:
            bad.append((key, 'missing', 0))
            continue
        v = flat[key]
        kind = key.rsplit('/', 1)[1]
        if kind == 'shape':
            if tuple(v) != tuple(g):
                bad.append((key, tuple(v), tuple(g)))
            continue
        base = key.rsplit('/', 1)[0]
        if kind in ('full', 'sample'):
            scale = float(np.abs(g).max()) if g.size else 0.0
        elif kind == 'mean':
            scale = float(golden[base + '/absmax'])
        else:
            scale = float(abs(g))
        is_grad = 'grads' in parts[0]
        rt = grad_rtol if (grad_rtol is not None and is_grad) else rtol
        if is_grad and grad_robust is not None and kind in ('l2', 'mean'):
            rt = min(rt, grad_robust[kind])    # the digests of a gradient tensor do not inherit the max-element tolerance
        tol = rt * max(scale, 1e-30) * atol_scale + 1e-12
        diff = np.abs(np.asarray(v, np.float64) - np.asarray(g, np.float64)) if g.size else np.zeros(1)
        err = float(diff.max())
        if is_grad and np.isfinite(err):
            what = 'grad_' + ('max' if kind in ('full', 'sample') else kind)
            if err / max(scale, 1e-30) >= rep.get(what, (-1.0, ''))[0]:
                rep[what] = (err / max(scale, 1e-30), key)
        if is_grad and grad_robust is not None and kind in ('full', 'sample') and g.size > 1 and scale > 0 and np.isfinite(err):
            a, b_ = np.asarray(v, np.float64).ravel(), np.asarray(g, np.float64).ravel()
            na, nb = float(np.linalg.norm(a)), float(np.linalg.norm(b_))
            one_minus_cos = 1.0 - float(a @ b_) / max(na * nb, 1e-300)
            q99, q90 = float(np.quantile(diff, 0.99)) / scale, float(np.quantile(diff, 0.90)) / scale
            for what, val in (('grad_one_minus_cosine', one_minus_cos), ('grad_q99', q99), ('grad_q90', q90)):
                if val >= rep.get(what, (-1.0, ''))[0]:
                    rep[what] = (val, key)
            if one_minus_cos > grad_robust['cosine']:
                bad.append((key + '#1-cosine', one_minus_cos, grad_robust['cosine']))
            if g.size >= 100:              # below that the quantiles ARE the max, which the max-element rule owns
                if q99 > grad_robust['q99']:
                    bad.append((key + '#q99', q99, grad_robust['q99']))
                if q90 > grad_robust['q90']:
                    bad.append((key + '#q90', q90, grad_robust['q90']))
        if 'params' in parts[0] and np.isfinite(err):
            # Post-Adam weights: the first Adam steps move every weight by ~lr*sign(g); where g is
            # ~0 its sign is round-off, so isolated elements legitimately differ by up to 2*lr per
            # step.  Require: every element within PARAM_ABS, and >= 97 % of them within tol (>= 75 % for
            # tensors whose whole range is below ~10 lr, e.g. the tiny Mapping biases, where tol << lr and
            # every sign-ambiguous element counts as an outlier; a wrong update rule would move ~100 %).
            frac_ok = 0.03 if tol >= 1e-5 else 0.25
            if err <= PARAM_ABS and (kind not in ('full', 'sample') or float((diff > tol).mean()) <= frac_ok):
                continue
        worst = max(worst, err / max(scale, 1e-30))
        if not np.isfinite(err) or err > tol:
            bad.append((key, err, tol))
    for case, fr in dead_off.items():
        if float(np.mean(fr)) > 0.15:
            bad.append((case + '/<dead biases: mean fraction of elements off the reference>', float(np.mean(fr)), 0.15))
    return bad, worst


# --------------------------------------------------------------------------------------
# adapters
# --------------------------------------------------------------------------------------
class NativeAdapter(object):
    """Drives an implementation with the oracle/product calling convention (explicit noise=)."""

    def __init__(self, module, device='cpu', trainer_kwargs=None):
        import torch
        self.torch = torch
        self.m = module
        self.device = device
        self.trainer_kwargs = trainer_kwargs or {}

    def T(self, a):
        if a is None:
            return None
        return self.torch.as_tensor(np.ascontiguousarray(a)).to(self.device)

    def N(self, t):
        return t.detach().cpu().numpy().copy()

    def make_trainer(self, hp, sds):
        tr = self.m.make_trainer(hp, self.device, **self.trainer_kwargs)
        for net in ('gen', 'dis', 'vae', 'map'):
            getattr(tr, net).load_state_dict({k: self.torch.as_tensor(v) for k, v in sds[net].items()})
        return tr

    def set_train(self, tr, flag):
        self.m.set_training(tr.gen, flag)

    def gen_forward(self, tr, xa, xb, nz):
        return [self.N(t) for t in tr.gen.forward(self.T(xa), self.T(xb), noise=self.T(nz))]

    def gen_encode(self, tr, xa, xb, na, nb):
        return [self.N(t) for t in tr.gen.encode(self.T(xa), self.T(xb), noise_a=self.T(na), noise_b=self.T(nb))]

    def gen_decode(self, tr, z):
        return [self.N(t) for t in tr.gen.decode(self.T(z))]

    def gen_a2b(self, tr, x, nz):
        return [self.N(t) for t in tr.gen.forward_a2b(self.T(x), noise=self.T(nz))]

    def gen_b2a(self, tr, x, nz):
        return [self.N(t) for t in tr.gen.forward_b2a(self.T(x), noise=self.T(nz))]

    def dis_forward(self, tr, xa, xb):
        return [self.N(t) for t in tr.dis.forward(self.T(xa), self.T(xb))]

    def dis_regress(self, tr, which, x):
        f = tr.dis.regress_a if which == 'a' else tr.dis.regress_b
        return [self.N(t) for t in f(self.T(x))]

    def dis_feats(self, tr, a, b, c, d):
        return [self.N(t) for t in tr.dis.feats(self.T(a), self.T(b), self.T(c), self.T(d))]

    def vae_forward(self, tr, y, nz):
        return [self.N(t) for t in tr.vae.forward(self.T(y), noise=self.T(nz))]

    def vae_decode(self, tr, z):
        return self.N(tr.vae.decode(self.T(z)))

    def map_forward(self, tr, z):
        return self.N(tr.map.forward(self.T(z)))

    def dis_update(self, tr, b, hp, nz):
        nz = tuple(self.T(n) for n in nz) if isinstance(nz, (tuple, list)) else self.T(nz)
        tr.dis_update(self.T(b['xa']), self.T(b['la']), self.T(b['xb']), self.T(b['lb']), self.T(b['ca']),
                      self.T(b['cb']), hp, noise=nz)

    def gen_update(self, tr, b, hp, nz3):
        out = tr.gen_update(self.T(b['xa']), self.T(b['la']), self.T(b['xb']), self.T(b['lb']), hp,
                            noise=tuple(self.T(n) for n in nz3))
        return [self.N(t) for t in out]

    def post_update(self, tr, b, mode, hp, nz_gen, nz_va, nz_vb):
        out = tr.post_update(self.T(b['xa']), self.T(b['la']), self.T(b['xb']), self.T(b['lb']), self.T(b['ca']),
                             self.T(b['cb']), mode, hp,
                             noise=dict(gen=self.T(nz_gen), vae_a=self.T(nz_va), vae_b=self.T(nz_vb)))
        return [self.N(t) for t in out]

    def vae_update(self, tr, y, hp, nz):
        return self.N(tr.vae_update(self.T(y), hp, noise=self.T(nz)))

    def scalars(self, tr):
        return {k: np.float64(np.asarray(getattr(tr, k)).reshape(-1)[0]) for k in sorted(vars(tr))
                if ('loss' in k or 'acc' in k) and not callable(getattr(tr, k)) and not k.endswith('criterion')
                and '_criterion' not in k}

    def params(self, tr, net):
        return OrderedDict((k, self.N(v)) for k, v in getattr(tr, net).state_dict().items())

    def grads(self, tr, net):
        return self.m.named_grads(getattr(tr, net), self.N)


# --------------------------------------------------------------------------------------
# the cases
# --------------------------------------------------------------------------------------
def make_inputs(n, label_dim=108):
    xa, la, ca = synth.make_batch(n, synth.YAML_SEED, label_dim)
    xb, lb, cb = synth.make_batch(n, synth.YAML_SEED + 1, label_dim)
    return dict(xa=xa, la=la, ca=ca, xb=xb, lb=lb, cb=cb)


def make_weights(hp, shapes_mod):
    """shapes_mod provides gen_shapes/dis_shapes/vae_shapes/map_shapes (key -> shape)."""
    return dict(gen=synth.make_state_dict(shapes_mod.gen_shapes(hp['gen']), 1),
                dis=synth.make_state_dict(shapes_mod.dis_shapes(hp['dis']), 2),
                vae=synth.make_state_dict(shapes_mod.vae_shapes(hp['vae']), 3),
                map=synth.make_state_dict(shapes_mod.map_shapes(hp['map']), 4))


def latent_shape(hp, n):
    c = hp['gen']['ch'] * 2 ** (hp['gen']['n_enc_front_blk'] - 1)
    s = 128 // 2 ** (hp['gen']['n_enc_front_blk'] - 1)
    return (n, c, s, s)


def run_module_cases(A, config, shapes_mod, with_map=True):
    """Forward-only cases (SURVEY §8(c) golden items 1, 2, 5)."""
    hp = hp_for(config)
    sds = make_weights(hp, shapes_mod)
    tr = A.make_trainer(hp, sds)
    n = 2
    b = make_inputs(n)
    R = OrderedDict()
    names5 = ('x_aa', 'x_ba', 'x_ab', 'x_bb', 'shared')

    A.set_train(tr, False)
    R['gen.forward.eval'] = OrderedDict(zip(names5, A.gen_forward(tr, b['xa'], b['xb'], None)))
    R['gen.encode.eval'] = OrderedDict(zip(('z_a', 'z_b'), A.gen_encode(tr, b['xa'], b['xb'], None, None)))
    z = noise(latent_shape(hp, 2 * n), 77, 0.5)
    R['gen.decode'] = OrderedDict(zip(('out_a', 'out_b'), A.gen_decode(tr, z)))
    R['gen.a2b.eval'] = OrderedDict(zip(('out', 'shared'), A.gen_a2b(tr, b['xa'], None)))
    R['gen.b2a.eval'] = OrderedDict(zip(('out', 'shared'), A.gen_b2a(tr, b['xb'], None)))
    A.set_train(tr, True)
    nz = noise(latent_shape(hp, 2 * n), 101)
    R['gen.forward.train'] = OrderedDict(zip(names5, A.gen_forward(tr, b['xa'], b['xb'], nz)))
    A.set_train(tr, False)

    R['dis.forward'] = OrderedDict(zip(('out_a', 'out_b', 'feats_a', 'feats_b'), A.dis_forward(tr, b['xa'], b['xb'])))
    R['dis.regress_a'] = OrderedDict(post=A.dis_regress(tr, 'a', b['xa'])[1])
    R['dis.regress_b'] = OrderedDict(post=A.dis_regress(tr, 'b', b['xb'])[1])
    R['dis.regress_b.n1'] = OrderedDict(post=A.dis_regress(tr, 'b', b['xb'][0:1])[1])      # squeeze() => [20]
    xs = [b['xa'], b['xb'], b['xb'][::-1].copy(), b['xa'][::-1].copy()]
    R['dis.feats'] = OrderedDict(zip(('f_aa', 'f_ba', 'f_ab', 'f_bb'), A.dis_feats(tr, *xs)))

    vn = noise((n, hp['vae']['z_dim']), 303, 0.05)
    R['vae.forward'] = OrderedDict(zip(('recons', 'z', 'mu', 'sd'), A.vae_forward(tr, b['la'], vn)))
    zp = noise((n, hp['vae']['z_dim']), 304, 0.3)
    R['vae.decode'] = OrderedDict(pose=A.vae_decode(tr, zp))
    if with_map and config == 'tiny':
        R['map.forward'] = OrderedDict(out=A.map_forward(tr, zp))
    return R


def run_resx_cases(A, shapes_mod):
    """SharedResXGen (ResNeXt generator, lsps_nets.py:277-387; unused by the shipped configs): forward in eval
    and one pretrain iteration, tiny width, k=2, cardinality 4."""
    hp = hp_for('tiny')
    hp['gen'] = dict(hp['gen'], name='SharedResXGen', n_resnext_k=2, n_resnext_c=4)
    sds = make_weights(hp, shapes_mod)
    tr = A.make_trainer(hp, sds)
    n = 2
    b = make_inputs(n)
    R = OrderedDict()
    A.set_train(tr, False)
    R['resx.forward.eval'] = OrderedDict(zip(('x_aa', 'x_ba', 'x_ab', 'x_bb', 'shared'),
                                             A.gen_forward(tr, b['xa'], b['xb'], None)))
    A.set_train(tr, True)
    lat2, lat1 = latent_shape(hp, 2 * n), latent_shape(hp, n)
    A.dis_update(tr, b, hp, noise(lat2, 1100))
    A.gen_update(tr, b, hp, (noise(lat2, 1200), noise(lat1, 1300), noise(lat1, 1400)))
    R['resx.pretrain.scalars'] = A.scalars(tr)
    _grad_digest(R, 'resx.pretrain.gen_update.grads', A, tr, 'gen')
    return R


def _grad_digest(R, case, A, tr, net):
    g = A.grads(tr, net)
    R[case] = OrderedDict((k, v) for k, v in g.items() if v is not None)


def run_step_cases(A, config, shapes_mod, n=2, post_n=8):
    """Update-step cases (SURVEY §8(c) golden item 3): losses, grads, weights after 1 and 2 steps."""
    hp = hp_for(config)
    sds = make_weights(hp, shapes_mod)
    R = OrderedDict()
    lat2 = latent_shape(hp, 2 * n)
    lat1 = latent_shape(hp, n)

    # ---- pretrain: dis_update -> gen_update, two iterations (depth_train.py:152-160)
    tr = A.make_trainer(hp, sds)
    A.set_train(tr, True)
    b = make_inputs(n)
    for it in range(2):
        A.dis_update(tr, b, hp, noise(lat2, 1000 + it))
        R['pretrain.it%d.dis_update.scalars' % it] = A.scalars(tr)
        if it == 0:
            _grad_digest(R, 'pretrain.it0.dis_update.grads', A, tr, 'dis')
        outs = A.gen_update(tr, b, hp, (noise(lat2, 2000 + it), noise(lat1, 3000 + it), noise(lat1, 4000 + it)))
        R['pretrain.it%d.gen_update.scalars' % it] = A.scalars(tr)
        if it == 0:
            _grad_digest(R, 'pretrain.it0.gen_update.grads', A, tr, 'gen')
            R['pretrain.it0.gen_update.outputs'] = OrderedDict(
                zip(('x_aa', 'x_ba', 'x_ab', 'x_bb', 'x_aba', 'x_bab'), outs[:6]))
        R['pretrain.it%d.dis.params' % it] = A.params(tr, 'dis')
        R['pretrain.it%d.gen.params' % it] = A.params(tr, 'gen')

    # ---- estimate modes 0 / 3 / 4 (post_update), batch post_n so the [0:4] slice matters
    bp = make_inputs(post_n)
    latp = latent_shape(hp, 8)
    zd = hp['vae']['z_dim']
    for mode in (0, 3, 4):
        tr = A.make_trainer(hp, sds)
        A.set_train(tr, True)
        for it in range(2):
            A.post_update(tr, bp, mode, hp, noise(latp, 5000 + it), noise((post_n, zd), 6000 + it, 0.05),
                          noise((post_n, zd), 7000 + it, 0.05))
            R['estimate%d.it%d.scalars' % (mode, it)] = A.scalars(tr)
            if it == 0:
                _grad_digest(R, 'estimate%d.it0.grads' % mode, A, tr, 'dis')
            R['estimate%d.it%d.dis.params' % (mode, it)] = A.params(tr, 'dis')

    # ---- pretrain with the Mapping branch (train_map: True, lsps_trainer.py:84-100,147-158,201-204)
    hpm = copy.deepcopy(hp)
    hpm['train_map'] = True
    tr = A.make_trainer(hpm, sds)
    A.set_train(tr, True)
    for it in range(2):
        A.dis_update(tr, b, hpm, (noise(lat2, 9000 + it), noise((2 * n, zd), 9100 + it, 0.05)))
        R['pretrain_map.it%d.dis_update.scalars' % it] = A.scalars(tr)
        outs = A.gen_update(tr, b, hpm, (noise(lat2, 9200 + it), noise(lat1, 9300 + it), noise(lat1, 9400 + it),
                                         noise((2 * n, zd), 9500 + it, 0.05)))
        R['pretrain_map.it%d.gen_update.scalars' % it] = A.scalars(tr)
        if it == 0:
            _grad_digest(R, 'pretrain_map.it0.gen_update.grads.gen', A, tr, 'gen')
            _grad_digest(R, 'pretrain_map.it0.gen_update.grads.map', A, tr, 'map')
            R['pretrain_map.it0.gen_update.outputs'] = OrderedDict(zip(('decode_A', 'decode_B'), outs[6:8]))
        R['pretrain_map.it%d.dis.params' % it] = A.params(tr, 'dis')
        R['pretrain_map.it%d.map.params' % it] = A.params(tr, 'map')

    # ---- stage-1 VAE step (lsps_trainer.py:62-74)
    tr = A.make_trainer(hp, sds)
    for it in range(2):
        dec = A.vae_update(tr, bp['la'], hp, noise((post_n, zd), 8000 + it, 0.05))
        R['vae_update.it%d' % it] = OrderedDict(dec=dec, **A.scalars(tr))
    R['vae_update.params'] = A.params(tr, 'vae')
    return R


def run_extra_cases(A, shapes_mod, which=('estimate1', 'wide')):
    """Round-2 additions (golden_extra.npz; the round-1 files stay byte-identical):
      * `post_update(mode=1)` — regress_b only (lsps_trainer.py:231-234), tiny and full width, two iterations;
      * one pretrain iteration at FULL width with 3 samples per domain: the generator's residual blocks then run on 6
        samples, the smallest batch at which the library's default ('auto') dispatch picks the Winograd kernels, so this
        is a trainer-level golden on the default dispatch of the bench."""
    R = OrderedDict()
    if 'estimate1' in which:
        for config in ('tiny', 'full'):
            hp = hp_for(config)
            sds = make_weights(hp, shapes_mod)
            post_n, zd = 8, hp['vae']['z_dim']
            bp = make_inputs(post_n)
            tr = A.make_trainer(hp, sds)
            A.set_train(tr, True)
            for it in range(2):
                A.post_update(tr, bp, 1, hp, None, None, noise((post_n, zd), 7100 + it, 0.05))
                R['%s.estimate1.it%d.scalars' % (config, it)] = A.scalars(tr)
                if it == 0:
                    _grad_digest(R, '%s.estimate1.it0.grads' % config, A, tr, 'dis')
                R['%s.estimate1.it%d.dis.params' % (config, it)] = A.params(tr, 'dis')
    if 'wide' in which:
        hp = hp_for('full')
        sds = make_weights(hp, shapes_mod)
        n = 3
        lat2, lat1 = latent_shape(hp, 2 * n), latent_shape(hp, n)
        tr = A.make_trainer(hp, sds)
        A.set_train(tr, True)
        b = make_inputs(n)
        A.dis_update(tr, b, hp, noise(lat2, 1500))
        R['wide.it0.dis_update.scalars'] = A.scalars(tr)
        _grad_digest(R, 'wide.it0.dis_update.grads', A, tr, 'dis')
        outs = A.gen_update(tr, b, hp, (noise(lat2, 2500), noise(lat1, 3500), noise(lat1, 4500)))
        R['wide.it0.gen_update.scalars'] = A.scalars(tr)
        _grad_digest(R, 'wide.it0.gen_update.grads', A, tr, 'gen')
        R['wide.it0.gen_update.outputs'] = OrderedDict(zip(('x_aa', 'x_ba', 'x_ab', 'x_bb', 'x_aba', 'x_bab'), outs[:6]))
        R['wide.it0.dis.params'] = A.params(tr, 'dis')
        R['wide.it0.gen.params'] = A.params(tr, 'gen')
    return R


def run_n16_cases(A, shapes_mod, n=16):
    """Round 5: one pretrain iteration and one estimate3 step at FULL width with 16 samples per domain —
    the smallest batch at which the product's DEFAULT dispatch is the bench's multi-round, XCD-mapped one (32-image residual
    convs on the F(4x4,3x3) kernels, 48-image discriminator passes).  No golden file: the test runs the CPU oracle on the same
    seeded inputs (about half a minute on the GPU box's host) and compares the HIP trainer with it."""
    hp = hp_for('full')
    sds = make_weights(hp, shapes_mod)
    R = OrderedDict()
    lat2, lat1 = latent_shape(hp, 2 * n), latent_shape(hp, n)
    tr = A.make_trainer(hp, sds)
    A.set_train(tr, True)
    b = make_inputs(n)
    A.dis_update(tr, b, hp, noise(lat2, 1600))
    R['n16.dis_update.scalars'] = A.scalars(tr)
    _grad_digest(R, 'n16.dis_update.grads', A, tr, 'dis')
    outs = A.gen_update(tr, b, hp, (noise(lat2, 2600), noise(lat1, 3600), noise(lat1, 4600)))
    R['n16.gen_update.scalars'] = A.scalars(tr)
    _grad_digest(R, 'n16.gen_update.grads', A, tr, 'gen')
    R['n16.gen_update.outputs'] = OrderedDict(zip(('x_aa', 'x_ba', 'x_ab', 'x_bb', 'x_aba', 'x_bab'), outs[:6]))
    zd = hp['vae']['z_dim']
    tr = A.make_trainer(hp, sds)
    A.set_train(tr, True)
    outs = A.post_update(tr, b, 3, hp, noise(latent_shape(hp, 8), 5600), noise((n, zd), 6600, 0.05), noise((n, zd), 7600, 0.05))
    R['n16.estimate3.scalars'] = A.scalars(tr)
    _grad_digest(R, 'n16.estimate3.grads', A, tr, 'dis')
    R['n16.estimate3.outputs'] = OrderedDict(zip(('x_aa', 'x_ba', 'x_ab', 'x_bb'), outs[:4]))
    return R


def run_trajectory(A, config, shapes_mod, n=4, n_pre=30, n_est=30, held_out=16, cadence=5, graphs=False, perturb=0.0):
    """a TRAINING TRAJECTORY, not two iterations.  The reference's loop (/root/reference/src/depth_train.py:140-166):
    `n_pre` pretrain iterations (`dis_update` -> `gen_update`, both schedulers stepped every `cadence` iterations: the driver's
    1000, scaled) and then `n_est` estimate iterations (`post_update(mode 3)`, the discriminator's scheduler every `cadence`: the
    driver's 100) on the SAME trainer, three different seeded batches in rotation, every random draw injected.  The schedulers
    are fast-forwarded to step 197 first (as after 197 000 iterations), so that MultiStepLR's first milestone (200: lr x 0.5,
    lsps_trainer.py:32-34) falls INSIDE the trajectory and the halved rate reaches Adam on both sides.  Then the A12 read-out
    (depth_train.py:200-253) of the trained regressor on `held_out` unseen samples.  Returns every loss / accuracy scalar per
    iteration, the learning rates seen, and the read-out.  `perturb`: relative noise on the initial weights (calibration of
    how fast two f32 trajectories drift apart by themselves)."""
    import warnings
    hp = hp_for(config)
    sds = make_weights(hp, shapes_mod)
    if perturb:
        rs = np.random.RandomState(77)
        sds = dict((net, OrderedDict((k, (v * (1.0 + perturb * rs.standard_normal(v.shape))).astype(v.dtype)) for k, v in sd.items()))
                   for net, sd in sds.items())
    tr = A.make_trainer(hp, sds)
    A.set_train(tr, True)
    if graphs:
        tr.use_graphs(True)
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')                 # "lr_scheduler.step() before optimizer.step()"
        for _ in range(197):
            tr.dis_sch.step()
            tr.gen_sch.step()
    R = OrderedDict()
    zd = hp['vae']['z_dim']
    lat2, lat1, lat8 = latent_shape(hp, 2 * n), latent_shape(hp, n), latent_shape(hp, 2 * min(4, n))
    batches = []
    for s in range(3):
        xa, la, ca = synth.make_batch(n, 5000 + 2 * s)
        xb, lb, cb = synth.make_batch(n, 5001 + 2 * s)
        batches.append(dict(xa=xa, la=la, ca=ca, xb=xb, lb=lb, cb=cb))
    lrs = []
    for it in range(n_pre):
        if (it + 1) % cadence == 0:                     # depth_train.py:154-156
            tr.dis_sch.step()
            tr.gen_sch.step()
        b = batches[it % 3]
        A.dis_update(tr, b, hp, noise(lat2, 11000 + it))
        A.gen_update(tr, b, hp, (noise(lat2, 12000 + it), noise(lat1, 13000 + it), noise(lat1, 14000 + it)))
        R['traj.pre.it%03d' % it] = A.scalars(tr)
        lrs.append((float(tr.dis_opt.param_groups[0]['lr']), float(tr.gen_opt.param_groups[0]['lr'])))
    for it in range(n_est):
        if (it + 1) % cadence == 0:                     # depth_train.py:163-164
            tr.dis_sch.step()
        b = batches[it % 3]
        A.post_update(tr, b, 3, hp, noise(lat8, 15000 + it), noise((n, zd), 16000 + it, 0.05), noise((n, zd), 17000 + it, 0.05))
        R['traj.est.it%03d' % it] = dict((k, v) for k, v in A.scalars(tr).items() if k.startswith('dis_'))
        lrs.append((float(tr.dis_opt.param_groups[0]['lr']), float(tr.gen_opt.param_groups[0]['lr'])))
    # A12 on held-out samples with the TRAINED regressor
    xb, lb, cb = synth.make_batch(held_out, 6001)
    getattr(tr.dis, 'eval', lambda: None)()        # the product's is an nn.Module; the oracle's net has no mode
    torch = A.torch
    with torch.no_grad():
        _, post, _ = tr.dis.regress_b(A.T(xb))
        pose = A.N(tr.vae.decode(post)).reshape(held_out, -1)
    cube = np.array([300.0, 300.0, 300.0], np.float32)
    idx = np.array([0, 3, 6, 9, 12, 15, 18, 21, 24, 25, 27, 30, 31, 32])
    gt = lb.reshape(held_out, -1, 3)[:, idx] * (cube[0] / 2.) + cb.reshape(held_out, 1, 3)
    pr = pose.reshape(held_out, -1, 3)[:, idx] * (cube[0] / 2.) + cb.reshape(held_out, 1, 3)
    err = np.sqrt(np.square(gt - pr).sum(axis=2))
    return dict(scalars=R, lrs=lrs, pose=pose, err=err, worst_joint=np.argmax(err, axis=1),
                mean_err=float(np.nanmean(np.nanmean(err, axis=1))), frames_within_40=int((np.nanmax(err, axis=1) <= 40).sum()),
                dis=A.params(tr, 'dis'))


def trajectory_envelope(ref, perturbed):
    """How far the REFERENCE's own trajectory moves when its initial weights are perturbed (`perturbed`: runs of the same oracle
    with `perturb=` 1e-5 .. 3e-5, the size of an f32 conv kernel's round-off): {scalar name: [running max over iterations of the
    relative deviation]}.  The L1 feature-matching term differentiates |f_b - f_a| (sign()), LeakyReLU masks flip, and Adam
    normalises every gradient, so a trajectory BIFURCATES after some tens of iterations whatever started the difference
    (profiles/r6c_trajectory_tiny.txt: the oracle against itself at 1e-5 leaves `dis_feat_loss` by 3e-3 from iteration 16)."""
    env = {}
    keys = list(ref['scalars'])
    for run in perturbed:
        for name in set(n for k in keys for n in ref['scalars'][k]):
            cur, series = 0.0, []
            for k in keys:
                if name in ref['scalars'][k]:
                    v = float(ref['scalars'][k][name])
                    cur = max(cur, abs(float(run['scalars'][k][name]) - v) / max(abs(v), 1e-30))
                series.append(cur)
            env[name] = [max(a, b) for a, b in zip(env.get(name, [0.0] * len(series)), series)]
    return env


def compare_trajectories(got, ref, rtol=1e-3, growth=10.0, envelope=None, env_factor=3.0):
    """Every scalar of iteration `it` (counted over both phases) within rtol * (1 + it / growth) of the reference's, relative to
    max(|reference|, the scalar's largest magnitude over the trajectory * 1e-2) — accuracies and vanishing losses are not held to a
    relative bound of their own tiny value.  With `envelope` (trajectory_envelope): a scalar outside that bound still passes when
    it is within `env_factor` x what the reference's OWN trajectory moves under a 1e-5 perturbation up to that iteration (a
    bifurcation, not an implementation error); those are returned as `chaotic`.
    Returns (failures, worst ratio of error to allowance, where[, chaotic])."""
    keys = list(ref['scalars'])
    assert list(got['scalars']) == keys
    scale = {}
    for k in keys:
        for name, v in ref['scalars'][k].items():
            scale[name] = max(scale.get(name, 0.0), abs(float(v)))
    bad, worst, where, chaotic = [], 0.0, None, []
    for it, k in enumerate(keys):
        allow = rtol * (1.0 + it / growth)
        for name, v in ref['scalars'][k].items():
            g = float(got['scalars'][k][name])
            if name.endswith('_acc'):
                # a count of >= 0.5 decisions over a handful of outputs of an untrained discriminator (they sit AT 0.5: dis_ad_loss
                # = 4 ln 2): one decision may legitimately fall the other way; more than one of the 2n would be a real difference
                r = abs(g - float(v)) / 0.26
            else:
                den = max(abs(float(v)), 1e-2 * scale[name], 1e-12)
                r = abs(g - float(v)) / den / allow
                if r > 1.0 and envelope is not None and np.isfinite(g):
                    e = env_factor * envelope.get(name, [0.0] * len(keys))[it]
                    if abs(g - float(v)) / den <= e:
                        chaotic.append((k, name, abs(g - float(v)) / den, e))
                        continue
            if r > worst:
                worst, where = r, (k, name, g, float(v))
            if r > 1.0 or not np.isfinite(g):
                bad.append((k, name, g, float(v)))
    if envelope is not None:
        return bad, worst, where, chaotic
    return bad, worst, where


def run_expand_cases(A, shapes_mod):
    """Round 4 (golden_expand.npz): `SharedDis` with the optional `n_expand_layer` key (lsps_nets.py:93,116-118): ONE stride-1
    3x3 LeakyReLUConv2d in front of the stride-2 trunk, tiny width (front 4 -> 8 channels, expand 8 -> 16 on 32 x 32, trunk
    16 -> 256).  Module outputs (forward incl. n = 1 `.squeeze()`, regress_b, feats) and two iterations of
    dis_update + post_update(mode 3) with gradients and post-Adam weights."""
    R = OrderedDict()
    hp = hp_for('tiny')
    hp['dis'] = dict(hp['dis'], n_expand_layer=1)
    sds = make_weights(hp, shapes_mod)
    tr = A.make_trainer(hp, sds)
    A.set_train(tr, False)
    b = make_inputs(2)
    R['expand.dis.forward'] = OrderedDict(zip(('out_a', 'out_b', 'feats_a', 'feats_b'), A.dis_forward(tr, b['xa'], b['xb'])))
    R['expand.dis.regress_b'] = OrderedDict(zip(('p0', 'p1', 'p2'), A.dis_regress(tr, 'b', b['xb'])))
    R['expand.dis.regress_a.n1'] = OrderedDict(zip(('p0', 'p1', 'p2'), A.dis_regress(tr, 'a', b['xa'][:1])))
    R['expand.dis.feats'] = OrderedDict(zip(('f0', 'f1', 'f2', 'f3'),
                                            A.dis_feats(tr, b['xa'][:1], b['xa'][1:], b['xb'][:1], b['xb'][1:])))
    n, post_n, zd = 2, 8, hp['vae']['z_dim']
    lat2 = latent_shape(hp, 2 * n)
    tr = A.make_trainer(hp, sds)
    A.set_train(tr, True)
    bp = make_inputs(post_n)
    for it in range(2):
        A.dis_update(tr, b, hp, noise(lat2, 8100 + it))
        R['expand.it%d.dis_update.scalars' % it] = A.scalars(tr)
        if it == 0:
            _grad_digest(R, 'expand.it0.dis_update.grads', A, tr, 'dis')
        A.post_update(tr, bp, 3, hp, noise(latent_shape(hp, 8), 8200 + it), noise((post_n, zd), 8300 + it, 0.05),
                      noise((post_n, zd), 8400 + it, 0.05))
        R['expand.it%d.estimate3.scalars' % it] = A.scalars(tr)
        if it == 0:
            _grad_digest(R, 'expand.it0.estimate3.grads', A, tr, 'dis')
        R['expand.it%d.dis.params' % it] = A.params(tr, 'dis')
    return R


# ---------------------------------------------------------------------------------------------
# residual block with dropout (`res_dropout_ratio` > 0: lsps_nets.py:176-179 -> common_net.py:171-172)
# ---------------------------------------------------------------------------------------------
DROP_P, DROP_CH, DROP_N, DROP_HW = 0.3, 16, 3, 32


def dropout_case_inputs():
    """Seeded input, weights, keep-mask/(1-p) and upstream gradient of the dropout residual-block case."""
    rs = np.random.RandomState(4242)
    shape = (DROP_N, DROP_CH, DROP_HW, DROP_HW)
    keep = (rs.uniform(size=shape) >= DROP_P).astype(np.float32) / np.float32(1.0 - DROP_P)
    return dict(x=noise(shape, 51), gy=noise(shape, 52), mask=keep,
                w0=noise((DROP_CH, DROP_CH, 3, 3), 53, 0.05), b0=noise((DROP_CH,), 54, 0.05),
                w3=noise((DROP_CH, DROP_CH, 3, 3), 55, 0.05), b3=noise((DROP_CH,), 56, 0.05))


# ---------------------------------------------------------------------------------------------
# the block classes of common_net.py that no shipped config instantiates (BatchNorm / ReLU / 2-D VAE variants)
# ---------------------------------------------------------------------------------------------
# name -> (constructor args, input shape)
BLOCK_CASES = OrderedDict([
    ('LeakyReLUBNConv2d', ((6, 10, 3, 2, 1), (5, 6, 12, 10))),
    ('LeakyReLUBNConvTranspose2d', ((6, 10, 3, 2, 1, 1), (5, 6, 7, 6))),
    ('LeakyReLUBNNSConv2d', ((6, 10, 3, 1, 1), (4, 6, 9, 8))),
    ('LeakyReLUBNNSConvTranspose2d', ((6, 10, 4, 2, 1), (4, 6, 5, 6))),
    ('LeakyReLUBNLinear', ((12, 20), (9, 12))),
    ('LeakyReLUBNNSResBlock', ((8, 8, 3, 1, 1), (4, 8, 10, 9))),
    ('INSResBlock', ((8, 8), (3, 8, 12, 12))),
    ('ReLUINSConv2d', ((6, 10, 3, 2, 1), (3, 6, 12, 10))),
    ('ReLUINSConvTranspose2d', ((6, 10, 3, 2, 1, 1), (3, 6, 7, 6))),
    ('LeakyReLUResBlock', ((8, 8, 3, 1, 1), (3, 8, 10, 9))),
    ('GaussianVAE2D', ((6, 5, 3, 1, 1), (3, 6, 8, 8))),
])


def run_block_case(mod, name, to_t, to_np):
    """Builds `mod.<name>`, loads seeded parameters, runs forward + backward in training mode (twice, so that running
    statistics move) and forward in eval mode.  Returns a flat dict of arrays."""
    import torch
    args, xshape = BLOCK_CASES[name]
    blk = getattr(mod, name)(*args)
    sd = blk.state_dict()
    new = {}
    for i, (k, v) in enumerate(sd.items()):
        if k.endswith('num_batches_tracked'):
            new[k] = v
        elif k.endswith('running_var'):
            new[k] = torch.as_tensor(np.abs(noise(tuple(v.shape), 700 + i, 0.3)) + 0.5)
        elif k.endswith('running_mean'):
            new[k] = torch.as_tensor(noise(tuple(v.shape), 700 + i, 0.2))
        else:
            new[k] = torch.as_tensor(noise(tuple(v.shape), 700 + i, 0.15 if v.dim() > 1 else 0.3))
    blk.load_state_dict(new)
    blk = to_t(blk)
    out = OrderedDict()
    blk.train()
    for it in range(2):
        x = to_t(torch.as_tensor(noise(xshape, 800 + it))).requires_grad_(True)
        for p in blk.parameters():
            p.grad = None
        y = blk(x)
        ys = y if isinstance(y, (tuple, list)) else (y,)
        loss = sum((yy * to_t(torch.as_tensor(noise(tuple(yy.shape), 900 + it + 10 * j)))).sum() for j, yy in enumerate(ys))
        loss.backward()
        for j, yy in enumerate(ys):
            out['%s/train%d/y%d' % (name, it, j)] = to_np(yy)
        out['%s/train%d/dx' % (name, it)] = to_np(x.grad)
        for k, p in blk.named_parameters():
            if p.grad is not None:
                out['%s/train%d/grad/%s' % (name, it, k)] = to_np(p.grad)
    for k, v in blk.state_dict().items():
        if 'running' in k:
            out['%s/buffers/%s' % (name, k)] = to_np(v)
    blk.eval()
    with torch.no_grad():
        y = blk(to_t(torch.as_tensor(noise(xshape, 810))))
        for j, yy in enumerate(y if isinstance(y, (tuple, list)) else (y,)):
            out['%s/eval/y%d' % (name, j)] = to_np(yy)
    return out
