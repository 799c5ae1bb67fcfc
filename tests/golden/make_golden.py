"""Generates tests/golden/golden_*.npz by running the REAL reference (imported from
/root/reference through ref_shim.py) on the seeded cases of cases.py.

Run in the build container only:  python tests/golden/make_golden.py [tiny|full|extra|expand|blocks|dropout|all]
The reference cannot travel to the GPU box; the committed .npz files are what pins the oracle
(tests/test_oracle_golden.py) and, through it and directly, the HIP path (tests/test_parity_gpu.py).
"""
import os
import sys
from collections import OrderedDict

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, REPO)
sys.path.insert(0, HERE)

import cases      # noqa: E402
import ref_shim   # noqa: E402


class NoiseQueue(object):
    """Replaces torch.randn / torch.normal inside the reference by a queue of recorded tensors
    (sites: common_net.py:39, lsps_nets.py:77)."""

    def __init__(self, torch):
        self.torch = torch
        self.q = []
        self._randn, self._normal = torch.randn, torch.normal

    def __enter__(self):
        t = self.torch

        def randn(*size, **kw):
            shape = tuple(size[0]) if len(size) == 1 and not isinstance(size[0], int) else tuple(size)
            nz = self.q.pop(0)
            assert tuple(nz.shape) == shape, (tuple(nz.shape), shape)
            return t.as_tensor(nz)

        def normal(mean, std=1.0, **kw):
            nz = self.q.pop(0)           # recorded tensors already carry the std
            assert tuple(nz.shape) == tuple(mean.shape), (tuple(nz.shape), tuple(mean.shape))
            return t.as_tensor(nz)

        t.randn, t.normal = randn, normal
        return self

    def __exit__(self, *a):
        self.torch.randn, self.torch.normal = self._randn, self._normal
        assert not self.q, "unused noise: %d" % len(self.q)


class RefAdapter(object):
    def __init__(self):
        self.ref = ref_shim.load_reference_trainers()
        import torch
        self.torch = torch

    def T(self, a):
        return None if a is None else self.torch.as_tensor(np.ascontiguousarray(a))

    def N(self, t):
        return t.detach().cpu().numpy().copy()

    def _call(self, fn, args, nzs):
        with NoiseQueue(self.torch) as q:
            q.q = [n for n in nzs if n is not None]
            out = fn(*args)
        return out

    def make_trainer(self, hp, sds):
        tr = self.ref.LSPSTrainer(hp)
        tr.cuda(0)
        for net in ('gen', 'dis', 'vae', 'map'):
            getattr(tr, net).load_state_dict({k: self.torch.as_tensor(v) for k, v in sds[net].items()}, strict=True)
        return tr

    def set_train(self, tr, flag):
        tr.gen.train(flag)

    def gen_forward(self, tr, xa, xb, nz):
        return [self.N(t) for t in self._call(tr.gen, (self.T(xa), self.T(xb)), [nz])]

    def gen_encode(self, tr, xa, xb, na, nb):
        return [self.N(t) for t in self._call(tr.gen.encode, (self.T(xa), self.T(xb)), [na, nb])]

    def gen_decode(self, tr, z):
        return [self.N(t) for t in tr.gen.decode(self.T(z))]

    def gen_a2b(self, tr, x, nz):
        return [self.N(t) for t in self._call(tr.gen.forward_a2b, (self.T(x),), [nz])]

    def gen_b2a(self, tr, x, nz):
        return [self.N(t) for t in self._call(tr.gen.forward_b2a, (self.T(x),), [nz])]

    def dis_forward(self, tr, xa, xb):
        return [self.N(t) for t in tr.dis(self.T(xa), self.T(xb))]

    def dis_regress(self, tr, which, x):
        f = tr.dis.regress_a if which == 'a' else tr.dis.regress_b
        return [self.N(t) for t in f(self.T(x))]

    def dis_feats(self, tr, a, b, c, d):
        return [self.N(t) for t in tr.dis.feats(self.T(a), self.T(b), self.T(c), self.T(d))]

    def vae_forward(self, tr, y, nz):
        return [self.N(t) for t in self._call(tr.vae, (self.T(y),), [nz])]

    def vae_decode(self, tr, z):
        return self.N(tr.vae.decode(self.T(z)))

    def map_forward(self, tr, z):
        return self.N(tr.map(self.T(z)))

    def dis_update(self, tr, b, hp, nz):
        self._call(tr.dis_update, (self.T(b['xa']), self.T(b['la']), self.T(b['xb']), self.T(b['lb']),
                                   self.T(b['ca']), self.T(b['cb']), hp), list(nz) if isinstance(nz, (tuple, list)) else [nz])

    def gen_update(self, tr, b, hp, nz3):
        out = self._call(tr.gen_update, (self.T(b['xa']), self.T(b['la']), self.T(b['xb']), self.T(b['lb']), hp),
                         list(nz3))
        return [self.N(t) for t in out]

    def post_update(self, tr, b, mode, hp, nz_gen, nz_va, nz_vb):
        # draw order inside the reference (lsps_trainer.py:227-252): gen noise, vae(a), vae(b)
        nzs = {0: [nz_va], 1: [nz_vb], 3: [nz_gen, nz_va], 4: [nz_gen, nz_va, nz_vb]}[mode]
        out = self._call(tr.post_update, (self.T(b['xa']), self.T(b['la']), self.T(b['xb']), self.T(b['lb']),
                                          self.T(b['ca']), self.T(b['cb']), mode, hp), nzs)
        return [self.N(t) for t in out]

    def vae_update(self, tr, y, hp, nz):
        return self.N(self._call(tr.vae_update, (self.T(y), hp), [nz]))

    def scalars(self, tr):
        out = {}
        for k in sorted(dir(tr)):
            if k.startswith('_') or not ('loss' in k or 'acc' in k):
                continue
            v = getattr(tr, k)
            if callable(v):
                continue
            out[k] = np.float64(np.asarray(v).reshape(-1)[0])
        return out

    def params(self, tr, net):
        return OrderedDict((k, self.N(v)) for k, v in getattr(tr, net).state_dict().items())

    def grads(self, tr, net):
        return OrderedDict((k, None if p.grad is None else self.N(p.grad))
                           for k, p in getattr(tr, net).named_parameters())


class RefShapes(object):
    """key -> shape tables read off the reference's own modules (not from the oracle)."""

    def __init__(self, ref):
        self.ref = ref

    def _shapes(self, cls, cfg):
        m = getattr(self.ref, cls)(cfg)
        return OrderedDict((k, tuple(v.shape)) for k, v in m.state_dict().items())

    def gen_shapes(self, cfg):
        return self._shapes(cfg['name'], cfg)

    def dis_shapes(self, cfg):
        return self._shapes(cfg['name'], cfg)

    def vae_shapes(self, cfg):
        return self._shapes(cfg['name'], cfg)

    def map_shapes(self, cfg):
        return self._shapes(cfg['name'], cfg)


def run_dropout_case(ref):
    """The reference's own LeakyINSResBlock(dropout=p) in training mode, with nn.Dropout's random keep mask replaced
    by the recorded one (F.dropout is patched for the duration), forward + backward; and in eval mode."""
    import torch
    import torch.nn.functional as F
    d = cases.dropout_case_inputs()
    blk = ref.LeakyINSResBlock(cases.DROP_CH, cases.DROP_CH, dropout=cases.DROP_P)
    assert type(blk.model[5]).__name__ == 'Dropout'
    blk.load_state_dict({'model.0.weight': torch.as_tensor(d['w0']), 'model.0.bias': torch.as_tensor(d['b0']),
                         'model.3.weight': torch.as_tensor(d['w3']), 'model.3.bias': torch.as_tensor(d['b3'])})
    mask = torch.as_tensor(d['mask'])
    orig = F.dropout

    def recorded(input, p=0.5, training=True, inplace=False):
        assert abs(p - cases.DROP_P) < 1e-12
        return input * mask if training else input
    F.dropout = recorded
    try:
        blk.train()
        x = torch.as_tensor(d['x']).clone().requires_grad_(True)
        y = blk(x)
        y.backward(torch.as_tensor(d['gy']))
        out = {'drop.train.y': y.detach().numpy().copy(), 'drop.train.dx': x.grad.numpy().copy(),
               'drop.train.dw0': blk.model[0].weight.grad.numpy().copy(),
               'drop.train.dw3': blk.model[3].weight.grad.numpy().copy()}
        blk.eval()
        with torch.no_grad():
            out['drop.eval.y'] = blk(torch.as_tensor(d['x']).clone()).numpy().copy()
    finally:
        F.dropout = orig
    return out


def main(which):
    import torch
    torch.set_num_threads(8)
    A = RefAdapter()
    if which in ('all', 'blocks'):
        R = OrderedDict()
        for name in cases.BLOCK_CASES:
            R.update(cases.run_block_case(A.ref, name, lambda t: t, lambda t: t.detach().cpu().numpy().copy()))
        path = os.path.join(HERE, 'golden_blocks.npz')
        np.savez_compressed(path, **R)
        print('block cases:', len(R), 'arrays ->', path, os.path.getsize(path) // 1024, 'KiB')
        if which == 'blocks':
            return
    if which in ('all', 'dropout'):
        path = os.path.join(HERE, 'golden_dropout.npz')
        np.savez_compressed(path, **run_dropout_case(A.ref))
        print('dropout case ->', path, os.path.getsize(path) // 1024, 'KiB')
        if which == 'dropout':
            return
    shapes = RefShapes(A.ref)
    if which in ('all', 'extra'):
        flat = cases.flatten(cases.run_extra_cases(A, shapes))
        path = os.path.join(HERE, 'golden_extra.npz')
        np.savez_compressed(path, **flat)
        print('extra', len(flat), 'arrays ->', path, os.path.getsize(path) // 1024, 'KiB')
        if which == 'extra':
            return
    if which in ('all', 'expand'):
        flat = cases.flatten(cases.run_expand_cases(A, shapes))
        path = os.path.join(HERE, 'golden_expand.npz')
        np.savez_compressed(path, **flat)
        print('expand', len(flat), 'arrays ->', path, os.path.getsize(path) // 1024, 'KiB')
        if which == 'expand':
            return
    for config in (['tiny', 'full'] if which == 'all' else [which]):
        R = OrderedDict()
        R.update(cases.run_module_cases(A, config, shapes))
        R.update(cases.run_step_cases(A, config, shapes))
        if config == 'tiny':
            R.update(cases.run_resx_cases(A, shapes))
        flat = cases.flatten(R)
        path = os.path.join(HERE, 'golden_%s.npz' % config)
        np.savez_compressed(path, **flat)
        print(config, len(flat), 'arrays ->', path, os.path.getsize(path) // 1024, 'KiB')


if __name__ == '__main__':
    main(sys.argv[1] if len(sys.argv) > 1 else 'all')
