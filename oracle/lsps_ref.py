"""ORACLE — test infrastructure only.  NOT part of the product path.

A CPU restatement (plain torch fp32 CPU ops, Python 3) of the reference's depth
encoder/decoder path: the four nets of ``src/trainers/lsps_nets.py``, the blocks of
``src/trainers/common_net.py`` that the shipped configs instantiate, the loss helpers and
the four update steps of ``src/trainers/lsps_trainer.py``.  It is written functionally over
a flat ``name -> tensor`` parameter table whose names ARE the reference's state-dict keys,
so reference checkpoints / golden weights load without translation.

Who may import this file: ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline``
leg of ``bench.py`` — as the checker / the timed CPU baseline, never as the thing shipped.
``lsps_amd`` never imports it.

Parity pin: PINNED against the reference itself.  The reference has no tests or golden
vectors of its own, so ``tests/golden/make_golden.py`` imports the real
``/root/reference/src/trainers`` in the build container (via ``tests/golden/ref_shim.py``),
runs it on seeded inputs with recorded noise, and commits the outputs under
``tests/golden/*.npz``; ``tests/test_oracle_golden.py`` checks this restatement against them.

Noise: the reference draws ``torch.randn`` (GaussianNoiseLayer, common_net.py:39) and
``torch.normal(std=0.05)`` (poseVAE.encode, lsps_nets.py:77) internally; here every such
site takes an explicit ``noise`` tensor (``None`` => draw from torch's global RNG).
"""
from collections import OrderedDict

import numpy as np
import torch
import torch.nn.functional as F

LRELU_SLOPE = 0.01     # nn.LeakyReLU() default, common_net.py:169,252,264
IN_EPS = 1e-5          # nn.InstanceNorm2d default eps, common_net.py:168


# --------------------------------------------------------------------------------------
# parameter tables
# --------------------------------------------------------------------------------------
class ParamTable(object):
    """Flat ``state-dict key -> leaf tensor`` table (replaces nn.Module in the oracle)."""

    def __init__(self, shapes):
        self.p = OrderedDict((k, torch.zeros(*s, dtype=torch.float32, requires_grad=True))
                             for k, s in shapes.items())

    def parameters(self):
        return list(self.p.values())

    def state_dict(self):
        return OrderedDict((k, v.detach().clone()) for k, v in self.p.items())

    def load_state_dict(self, sd):
        missing = [k for k in self.p if k not in sd]
        if missing:
            raise KeyError("missing keys: %s" % missing[:4])
        with torch.no_grad():
            for k, v in self.p.items():
                v.copy_(torch.as_tensor(np.asarray(sd[k]), dtype=torch.float32).reshape(v.shape))

    def zero_grad(self):
        for v in self.p.values():
            v.grad = None


def _res_block_shapes(s, prefix, t, cfg):
    """Residual block parameters: LeakyINSResBlock (common_net.py:160-175) or, for SharedResXGen,
    LeakyINSResNeXtBlock (common_net.py:111-126: 1x1 expand, grouped 3x3, 1x1 project)."""
    if cfg.get('name') == 'SharedResXGen':
        k, g = cfg.get('n_resnext_k', 1), cfg.get('n_resnext_c', 4)
        s[prefix + '.model.0.weight'], s[prefix + '.model.0.bias'] = (k * t, t, 1, 1), (k * t,)
        s[prefix + '.model.3.weight'], s[prefix + '.model.3.bias'] = (k * t, k * t // g, 3, 3), (k * t,)
        s[prefix + '.model.6.weight'], s[prefix + '.model.6.bias'] = (t, k * t, 1, 1), (t,)
    else:
        for m in (0, 3):
            s['%s.model.%d.weight' % (prefix, m)] = (t, t, 3, 3)
            s['%s.model.%d.bias' % (prefix, m)] = (t,)


def gen_shapes(cfg):
    """Key/shape list of SharedResGen / SharedResXGen (lsps_nets.py:164-237, 277-355)."""
    ch = cfg['ch']
    s = OrderedDict()
    for d, cin in (('A', cfg['input_dim_a']), ('B', cfg['input_dim_b'])):
        i = 0
        s['encode_%s.%d.model.0.weight' % (d, i)] = (ch, cin, 7, 7)          # :186-187
        s['encode_%s.%d.model.0.bias' % (d, i)] = (ch,)
        t = ch
        for i in range(1, cfg['n_enc_front_blk']):                           # :189-192
            s['encode_%s.%d.model.0.weight' % (d, i)] = (2 * t, t, 3, 3)
            s['encode_%s.%d.model.0.bias' % (d, i)] = (2 * t,)
            t *= 2
        i = cfg['n_enc_front_blk']
        for j in range(cfg['n_enc_res_blk']):                                # :194-196
            _res_block_shapes(s, 'encode_%s.%d' % (d, i + j), t, cfg)
    tch = t
    for grp, n in (('enc_shared', cfg['n_enc_shared_blk']), ('dec_shared', cfg['n_gen_shared_blk'])):
        for j in range(n):                                                   # :203-209
            _res_block_shapes(s, '%s.%d' % (grp, j), tch, cfg)
    for d, cout in (('A', cfg['input_dim_a']), ('B', cfg['input_dim_b'])):
        t = tch
        i = 0
        for j in range(cfg['n_gen_res_blk']):                                # :218-220
            _res_block_shapes(s, 'decode_%s.%d' % (d, i), t, cfg)
            i += 1
        for j in range(1, cfg['n_gen_front_blk']):                           # :222-225 (ConvTranspose: C_in,C_out,R,S)
            s['decode_%s.%d.model.0.weight' % (d, i)] = (t, t // 2, 3, 3)
            s['decode_%s.%d.model.0.bias' % (d, i)] = (t // 2,)
            t //= 2
            i += 1
        s['decode_%s.%d.weight' % (d, i)] = (t, cout, 1, 1)                  # :226-227
        s['decode_%s.%d.bias' % (d, i)] = (cout,)
    # order the keys like the reference's module registration order
    order = ['encode_A', 'encode_B', 'enc_shared', 'dec_shared', 'decode_A', 'decode_B']
    out = OrderedDict()
    for o in order:
        for k, v in s.items():
            if k.startswith(o + '.'):
                out[k] = v
    return out


def dis_shapes(cfg):
    """Key/shape list of SharedDis (lsps_nets.py:86-126)."""
    ch = cfg['ch']
    s = OrderedDict()
    tch = ch
    for d, cin in (('A', cfg['input_dim_a']), ('B', cfg['input_dim_b'])):
        s['model_%s.0.model.0.weight' % d] = (ch, cin, 7, 7)                 # :104
        s['model_%s.0.model.0.bias' % d] = (ch,)
        tch = ch
        for i in range(1, cfg['n_front_layer']):                             # :106-108
            s['model_%s.%d.model.0.weight' % (d, i)] = (2 * tch, tch, 3, 3)
            s['model_%s.%d.model.0.bias' % (d, i)] = (2 * tch,)
            tch *= 2
    n_expand = cfg.get('n_expand_layer', 0)
    for i in range(n_expand + cfg['n_shared_layer']):                        # :116-121
        s['model_S.%d.model.0.weight' % i] = (2 * tch, tch, 3, 3)
        s['model_S.%d.model.0.bias' % i] = (2 * tch,)
        tch *= 2
    s['D.weight'] = (1, tch, 1, 1)                                           # :124
    s['D.bias'] = (1,)
    s['Post.weight'] = (cfg['post_dim'], tch, 2, 2)                          # :123
    s['Post.bias'] = (cfg['post_dim'],)
    return s


def vae_shapes(cfg):
    """poseVAE (lsps_nets.py:34-59)."""
    i, z, h = cfg['input_dim'], cfg['z_dim'], cfg['h_dim']
    return OrderedDict([
        ('en_fc1.weight', (h, i)), ('en_fc1.bias', (h,)),
        ('en_mu.weight', (z, h)), ('en_mu.bias', (z,)),
        ('en_sigma.weight', (z, h)), ('en_sigma.bias', (z,)),
        ('de_fc1.model.0.weight', (h, z)), ('de_fc1.model.0.bias', (h,)),
        ('de_fc2.weight', (i, h)), ('de_fc2.bias', (i,)),
    ])


def map_shapes(cfg):
    """Mapping (lsps_nets.py:8-25); ConvTranspose weights are (C_in, C_out, 4, 4)."""
    ch = cfg['output_ch']
    dims = [(cfg['input_dim'], 4 * ch), (4 * ch, 4 * ch), (4 * ch, 2 * ch)]
    s = OrderedDict()
    for i, (a, b) in enumerate(dims):
        s['model.%d.model.0.weight' % i] = (a, b, 4, 4)
        s['model.%d.model.0.bias' % i] = (b,)
    s['model.3.weight'] = (2 * ch, ch, 4, 4)
    s['model.3.bias'] = (ch,)
    return s


# --------------------------------------------------------------------------------------
# blocks (common_net.py)
# --------------------------------------------------------------------------------------
def lrelu_conv(x, p, key, stride, pad):
    """LeakyReLUConv2d (common_net.py:246-256)."""
    return F.leaky_relu(F.conv2d(x, p[key + '.weight'], p[key + '.bias'], stride=stride, padding=pad),
                        LRELU_SLOPE)


def lrelu_convT(x, p, key, stride, pad, outpad):
    """LeakyReLUConvTranspose2d (common_net.py:258-268)."""
    return F.leaky_relu(F.conv_transpose2d(x, p[key + '.weight'], p[key + '.bias'], stride=stride,
                                           padding=pad, output_padding=outpad), LRELU_SLOPE)


def instance_norm(x):
    """nn.InstanceNorm2d(affine=False, no running stats) — per-(n,c) biased variance,
    identical in train and eval (common_net.py:168,171)."""
    mu = x.mean(dim=(2, 3), keepdim=True)
    var = ((x - mu) ** 2).mean(dim=(2, 3), keepdim=True)
    return (x - mu) / torch.sqrt(var + IN_EPS)


def leaky_ins_res_block(x, p, key, drop_mask=None):
    """LeakyINSResBlock (common_net.py:160-181): x + [Dropout](IN(conv(LReLU(IN(conv(x)))))).  `drop_mask`: the
    keep mask divided by (1-p) that nn.Dropout(p) multiplies by in training mode (:171-172); None = no dropout."""
    h = F.conv2d(x, p[key + '.model.0.weight'], p[key + '.model.0.bias'], stride=1, padding=1)
    h = F.leaky_relu(instance_norm(h), LRELU_SLOPE)
    h = F.conv2d(h, p[key + '.model.3.weight'], p[key + '.model.3.bias'], stride=1, padding=1)
    h = instance_norm(h)
    if drop_mask is not None:
        h = h * drop_mask
    return x + h


def leaky_ins_resnext_block(x, p, key, groups):
    """LeakyINSResNeXtBlock (common_net.py:111-132)."""
    h = F.conv2d(x, p[key + '.model.0.weight'], p[key + '.model.0.bias'])
    h = F.leaky_relu(instance_norm(h), LRELU_SLOPE)
    h = F.conv2d(h, p[key + '.model.3.weight'], p[key + '.model.3.bias'], padding=1, groups=groups)
    h = F.leaky_relu(instance_norm(h), LRELU_SLOPE)
    h = F.conv2d(h, p[key + '.model.6.weight'], p[key + '.model.6.bias'])
    return x + instance_norm(h)


# --------------------------------------------------------------------------------------
# nets (lsps_nets.py)
# --------------------------------------------------------------------------------------
class RefGen(ParamTable):
    """SharedResGen (lsps_nets.py:164-272)."""

    def __init__(self, cfg):
        super(RefGen, self).__init__(gen_shapes(cfg))
        self.cfg = cfg
        self.training = True
        if cfg.get('name') == 'SharedResXGen':                    # lsps_nets.py:277-387
            g = cfg.get('n_resnext_c', 4)
            self._block = lambda h, p, key: leaky_ins_resnext_block(h, p, key, g)
        else:
            self._block = leaky_ins_res_block

    def _enc_front(self, d, x):                                  # encode_A / encode_B
        c, p = self.cfg, self.p
        h = lrelu_conv(x, p, 'encode_%s.0.model.0' % d, 1, 3)
        for i in range(1, c['n_enc_front_blk']):
            h = lrelu_conv(h, p, 'encode_%s.%d.model.0' % (d, i), 2, 1)
        for j in range(c['n_enc_res_blk']):
            h = self._block(h, p, 'encode_%s.%d' % (d, c['n_enc_front_blk'] + j))
        return h

    def _enc_shared(self, h, noise):                             # enc_shared (+ GaussianNoiseLayer, common_net.py:32-40)
        for j in range(self.cfg['n_enc_shared_blk']):
            h = self._block(h, self.p, 'enc_shared.%d' % j)
        if self.training:
            h = h + (torch.randn(h.shape) if noise is None else noise)
        return h

    def _dec_shared(self, h):
        for j in range(self.cfg['n_gen_shared_blk']):
            h = self._block(h, self.p, 'dec_shared.%d' % j)
        return h

    def _dec_front(self, d, h):                                  # decode_A / decode_B
        c, p = self.cfg, self.p
        i = 0
        for j in range(c['n_gen_res_blk']):
            h = self._block(h, p, 'decode_%s.%d' % (d, i))
            i += 1
        for j in range(1, c['n_gen_front_blk']):
            h = lrelu_convT(h, p, 'decode_%s.%d.model.0' % (d, i), 2, 1, 1)
            i += 1
        h = F.conv_transpose2d(h, p['decode_%s.%d.weight' % (d, i)], p['decode_%s.%d.bias' % (d, i)])
        return torch.tanh(h)

    def forward(self, x_A, x_B, noise=None):                     # :250-258
        out = torch.cat((self._enc_front('A', x_A), self._enc_front('B', x_B)), 0)
        shared = self._enc_shared(out, noise)
        out = self._dec_shared(shared)
        out_A, out_B = self._dec_front('A', out), self._dec_front('B', out)
        n = x_A.size(0)
        x_Aa, x_Ba = torch.split(out_A, n, dim=0)
        x_Ab, x_Bb = torch.split(out_B, n, dim=0)
        return x_Aa, x_Ba, x_Ab, x_Bb, shared

    __call__ = forward

    def encode(self, x_A, x_B, noise_a=None, noise_b=None):      # :245-248
        return (self._enc_shared(self._enc_front('A', x_A), noise_a),
                self._enc_shared(self._enc_front('B', x_B), noise_b))

    def decode(self, z):                                         # :239-243
        out = self._dec_shared(z)
        return self._dec_front('A', out), self._dec_front('B', out)

    def forward_a2b(self, x_A, noise=None):                      # :260-265
        shared = self._enc_shared(self._enc_front('A', x_A), noise)
        return self._dec_front('B', self._dec_shared(shared)), shared

    def forward_b2a(self, x_B, noise=None):                      # :267-272
        shared = self._enc_shared(self._enc_front('B', x_B), noise)
        return self._dec_front('A', self._dec_shared(shared)), shared


class RefDis(ParamTable):
    """SharedDis (lsps_nets.py:86-160)."""

    def __init__(self, cfg):
        super(RefDis, self).__init__(dis_shapes(cfg))
        self.cfg = cfg
        self.n_expand = cfg.get('n_expand_layer', 0)                  # :93 optional key, 0 in both shipped configs
        self.n_shared = self.n_expand + cfg['n_shared_layer']

    def _front(self, d, x):                                      # :101-109
        h = lrelu_conv(x, self.p, 'model_%s.0.model.0' % d, 2, 3)
        for i in range(1, self.cfg['n_front_layer']):
            h = lrelu_conv(h, self.p, 'model_%s.%d.model.0' % (d, i), 2, 1)
        return h

    def _shared(self, h):                                        # :111-126
        for i in range(self.n_shared):                           # :116-118 expand layers: stride 1; :119-121 stride 2
            h = lrelu_conv(h, self.p, 'model_S.%d.model.0' % i, 1 if i < self.n_expand else 2, 1)
        return h

    def _post(self, f):
        post = F.conv2d(f, self.p['Post.weight'], self.p['Post.bias']).squeeze()   # :138-139 (squeeze: [20] at n=1)
        return post, post, post

    def regress_a(self, x_A):                                    # :135-139
        return self._post(self._shared(self._front('A', x_A)))

    def regress_b(self, x_B):                                    # :141-145
        return self._post(self._shared(self._front('B', x_B)))

    def feats(self, x_aa, x_ba, x_ab, x_bb):                     # :147-152
        f = torch.cat((self._front('A', torch.cat((x_aa, x_ba), 0)),
                       self._front('B', torch.cat((x_ab, x_bb), 0))), 0)
        f = self._shared(f)
        return torch.split(f, f.size(0) // 4, dim=0)

    def forward(self, x_A, x_B):                                 # :154-160
        f = self._shared(torch.cat((self._front('A', x_A), self._front('B', x_B)), 0))
        out_D = F.conv2d(f, self.p['D.weight'], self.p['D.bias'])
        f_A, f_B = torch.split(f, f.size(0) // 2, dim=0)
        o_A, o_B = torch.split(out_D, out_D.size(0) // 2, dim=0)
        return o_A.reshape(-1), o_B.reshape(-1), f_A, f_B

    __call__ = forward


class RefVAE(ParamTable):
    """poseVAE (lsps_nets.py:34-83)."""

    def __init__(self, cfg):
        super(RefVAE, self).__init__(vae_shapes(cfg))
        self.cfg = cfg

    def encode(self, y, noise=None):                             # :73-78 (noise ~ N(0, 0.05) ALWAYS)
        p = self.p
        h = F.leaky_relu(F.linear(y, p['en_fc1.weight'], p['en_fc1.bias']), LRELU_SLOPE)
        mu = F.linear(h, p['en_mu.weight'], p['en_mu.bias'])
        sd = F.softplus(F.linear(h, p['en_sigma.weight'], p['en_sigma.bias']))
        if noise is None:
            noise = torch.normal(torch.zeros(mu.size()), std=0.05)
        return mu + sd * noise, mu, sd

    def decode(self, z):                                         # :80-83
        p = self.p
        h = F.leaky_relu(F.linear(z, p['de_fc1.model.0.weight'], p['de_fc1.model.0.bias']), LRELU_SLOPE)
        return F.linear(h, p['de_fc2.weight'], p['de_fc2.bias'])

    def forward(self, y, noise=None):                            # :67-71
        z, mu, sd = self.encode(y, noise)
        return self.decode(z), z, mu, sd

    __call__ = forward


class RefMapping(ParamTable):
    """Mapping (lsps_nets.py:8-31)."""

    def __init__(self, cfg):
        super(RefMapping, self).__init__(map_shapes(cfg))

    def forward(self, x):
        h = x.unsqueeze(2).unsqueeze(3)
        h = lrelu_convT(h, self.p, 'model.0.model.0', 1, 0, 0)
        h = lrelu_convT(h, self.p, 'model.1.model.0', 2, 1, 0)
        h = lrelu_convT(h, self.p, 'model.2.model.0', 2, 1, 0)
        return F.conv_transpose2d(h, self.p['model.3.weight'], self.p['model.3.bias'], stride=2, padding=1)

    __call__ = forward


# --------------------------------------------------------------------------------------
# losses / accuracy (lsps_trainer.py:42-60, helpers.py:20-32)
# --------------------------------------------------------------------------------------
def l1_mean(a, b):
    return (a - b).abs().mean()                                  # nn.L1Loss(), lsps_trainer.py:44-49


def l2_mean(a, b):
    return ((a - b) ** 2).mean()                                 # :51-52


def kl(mu, sd=None):                                             # :55-60
    if sd is None:
        return (mu ** 2).mean()
    return (mu ** 2 + sd ** 2 - torch.log(sd ** 2)).sum() / mu.size(0)


def bce(prob, target_value):
    """F.binary_cross_entropy(prob, const target) with torch's log clamp at -100."""
    t = torch.full_like(prob, float(target_value))
    return F.binary_cross_entropy(prob, t)


def true_acc(prob):
    return float((prob.detach() >= 0.5).sum().item()) / (1.0 * prob.size(0))     # helpers.py:20-25


def fake_acc(prob):
    return float((prob.detach() <= 0.5).sum().item()) / (1.0 * prob.size(0))     # helpers.py:27-32


# --------------------------------------------------------------------------------------
# trainer (lsps_trainer.py)
# --------------------------------------------------------------------------------------
class RefTrainer(object):
    """LSPSTrainer restated (lsps_trainer.py:16-262).  ``literal=True`` keeps the reference's
    wasted backward scope (dis_update / post_update back-prop into gen; gen_update computes
    dis weight grads) so that CPU-baseline timings are not flattered; results are identical."""

    def __init__(self, hp, literal=True):
        self.hp = hp
        self.literal = literal
        lr = hp['lr']
        self.dis, self.gen = RefDis(hp['dis']), RefGen(hp['gen'])
        self.vae, self.map = RefVAE(hp['vae']), RefMapping(hp['map'])
        A = torch.optim.Adam                                                      # :26-29
        self.dis_opt = A(self.dis.parameters(), lr=lr, betas=(0.5, 0.999), weight_decay=0.0001)
        self.gen_opt = A(self.gen.parameters() + self.map.parameters(), lr=lr, betas=(0.5, 0.999),
                         weight_decay=0.0001)
        self.vae_opt = A(self.vae.parameters(), lr=lr * 10., betas=(0.5, 0.999), weight_decay=0.001)
        S = torch.optim.lr_scheduler.MultiStepLR                                  # :32-34
        self.dis_sch = S(self.dis_opt, milestones=[200, 300, 400, 450], gamma=0.5)
        self.gen_sch = S(self.gen_opt, milestones=[200, 300, 400, 450], gamma=0.5)
        self.vae_sch = S(self.vae_opt, milestones=[125, 175], gamma=0.1)

    # -- :62-74
    def vae_update(self, y, hp, noise=None):
        self.vae.zero_grad()
        dec, z, mu, sd = self.vae(y, noise)
        total = hp['kl_loss_vae'] * kl(mu, sd) + hp['ll_loss_vae'] * l1_mean(dec, y)
        total.backward()
        self.vae_opt.step()
        self.vae_total_loss = total.detach().numpy()
        return dec

    def _pose2depth(self, labels_a, labels_b, nz):
        """labels -> vae code -> Mapping -> gen.decode; halves as in lsps_trainer.py:87-93,148-154."""
        enc_pose, _, _ = self.vae.encode(torch.cat((labels_a, labels_b), 0), nz)
        z = self.map(enc_pose)
        dec_A, dec_B = self.gen.decode(z)
        half = dec_A.size(0) // 2
        return z, dec_A[:half], dec_B[half:]

    # -- :76-141 (noise = gen / a2b / b2a draws [+ vae draw when train_map])
    def gen_update(self, images_a, labels_a, images_b, labels_b, hp, noise=(None, None, None)):
        self.gen.zero_grad()
        x_aa, x_ba, x_ab, x_bb, shared = self.gen(images_a, images_b, noise[0])
        x_bab, shared_bab = self.gen.forward_a2b(x_ba, noise[1])
        x_aba, shared_aba = self.gen.forward_b2a(x_ab, noise[2])
        m_z = m_a = m_b = 0.
        decode_A, decode_B, data_a, data_b = x_ba, x_ab, x_ba, x_ab
        if hp['train_map']:                                                       # :84-100
            self.map.zero_grad()
            z_p2d, decode_A, decode_B = self._pose2depth(labels_a, labels_b, noise[3] if len(noise) > 3 else None)
            data_a, data_b = torch.cat((x_ba, decode_A), 0), torch.cat((x_ab, decode_B), 0)
            m_z = l2_mean(shared, z_p2d)
            m_a, m_b = l1_mean(decode_A, images_a), l1_mean(decode_B, images_b)
        outs_a, outs_b, _, _ = self.dis(data_a, data_b)
        ad_a, ad_b = bce(torch.sigmoid(outs_a), 1.0), bce(torch.sigmoid(outs_b), 1.0)
        enc = kl(shared)
        enc_bab, enc_aba = kl(shared_bab), kl(shared_aba)
        ll_a, ll_b = l1_mean(x_aa, images_a), l1_mean(x_bb, images_b)
        ll_aba, ll_bab = l1_mean(x_aba, images_a), l1_mean(x_bab, images_b)
        total = hp['gan_w'] * (ad_a + ad_b) + hp['ll_direct_link_w'] * (ll_a + ll_b) + \
            hp['ll_cycle_link_w'] * (ll_aba + ll_bab) + hp['kl_direct_link_w'] * (enc + enc) + \
            hp['kl_cycle_link_w'] * (enc_bab + enc_aba) + \
            hp['ll_map_z_w'] * m_z + hp['ll_map_w'] * (m_a + m_b)                 # :121-127 (enc doubled)
        if self.literal:
            total.backward()
        else:
            ps = self.gen.parameters() + (self.map.parameters() if hp['train_map'] else [])
            for p_, g in zip(ps, torch.autograd.grad(total, ps)):
                p_.grad = g
        self.gen_opt.step()
        self.gen_enc_loss = enc.detach().numpy()
        self.gen_enc_loss2 = (enc_aba + enc_bab).detach().numpy()
        self.gen_ad_loss = (ad_a + ad_b).detach().numpy()
        self.gen_ll_loss = (ll_a + ll_b).detach().numpy()
        self.gen_ll_loss2 = (ll_bab + ll_aba).detach().numpy()
        if hp['train_map']:
            self.gen_map_loss = m_z.detach().numpy()
            self.gen_map_loss2 = (m_a + m_b).detach().numpy()
        self.gen_total_loss = total.detach().numpy()
        return (x_aa, x_ba, x_ab, x_bb, x_aba, x_bab, decode_A, decode_B)

    # -- :143-218 ; noise = gen draw, or (gen draw, vae draw) when train_map
    def dis_update(self, images_a, labels_a, images_b, labels_b, com_a, com_b, hp, feat_mat=True, noise=None):
        self.dis.zero_grad()
        nz_gen, nz_vae = (noise if isinstance(noise, (tuple, list)) else (noise, None))
        import contextlib
        with (contextlib.nullcontext() if self.literal else torch.no_grad()):
            x_aa, x_ba, x_ab, x_bb, _ = self.gen(images_a, images_b, nz_gen)
            if hp['train_map']:
                _, decode_A, decode_B = self._pose2depth(labels_a, labels_b, nz_vae)
        if hp['train_map']:                                                       # :147-158
            data_a = torch.cat((images_a, x_ba, x_aa, decode_A), 0)
            data_b = torch.cat((images_b, x_ab, x_bb, decode_B), 0)
            ndiv = 4
        elif feat_mat:
            data_a, data_b, ndiv = torch.cat((images_a, x_ba, x_aa), 0), torch.cat((images_b, x_ab, x_bb), 0), 3
        else:
            data_a, data_b, ndiv = torch.cat((images_a, x_ba), 0), torch.cat((images_b, x_ab), 0), 2
        res_a, res_b, feats_a, feats_b = self.dis(data_a, data_b)
        fl_a = fl_b = 0.
        if feat_mat:                                                              # :171-177
            fa = torch.split(feats_a, feats_a.size(0) // ndiv, 0)
            fb = torch.split(feats_b, feats_a.size(0) // ndiv, 0)
            zero = torch.zeros_like(fa[2])
            fl_a, fl_b = l1_mean(fb[1] - fa[2], zero), l1_mean(fa[1] - fb[2], zero)
        oa = torch.split(torch.sigmoid(res_a), res_a.size(0) // ndiv, 0)
        ob = torch.split(torch.sigmoid(res_b), res_b.size(0) // ndiv, 0)
        ad_a = bce(oa[0], 1.0) + bce(oa[1], 0.0)                                  # :189-205
        ad_b = bce(ob[0], 1.0) + bce(ob[1], 0.0)
        if hp['train_map']:                                                       # :201-204
            ad_a = ad_a + bce(oa[3], 0.0)
            ad_b = ad_b + bce(ob[3], 0.0)
        self.dis_true_acc = 0.5 * (true_acc(oa[0]) + true_acc(ob[0]))
        self.dis_fake_acc = 0.5 * (fake_acc(oa[1]) + fake_acc(ob[1]))
        loss = hp['gan_w'] * (ad_a + ad_b) + hp['feature_w'] * (fl_a + fl_b)
        loss.backward()
        self.dis_opt.step()
        self.dis_ad_loss = (ad_a + ad_b).detach().numpy()
        if feat_mat:
            self.dis_feat_loss = (fl_a + fl_b).detach().numpy()
        self.dis_loss = loss.detach().numpy()

    # -- :220-262 ; noise = dict(gen=..., vae_a=..., vae_b=...)
    def post_update(self, images_a, labels_a, images_b, labels_b, com_a, com_b, mode, hp, noise=None):
        noise = noise or {}
        self.dis.zero_grad()
        x_aa, x_ba, x_ab, x_bb = images_a, images_a, images_b, images_b
        fl_a = fl_b = 0.
        rl_a = rl_b = 0.

        def reg(which, images, labels, nz):
            _, pred, _ = (self.dis.regress_a if which == 'a' else self.dis.regress_b)(images)
            target, _, _ = self.vae.encode(labels, nz)
            return l2_mean(pred, target if self.literal else target.detach())

        if mode == 0:
            rl_a = reg('a', images_a, labels_a, noise.get('vae_a'))
        elif mode == 1:
            rl_b = reg('b', images_b, labels_b, noise.get('vae_b'))
        else:
            if self.literal:
                x_aa, x_ba, x_ab, x_bb, _ = self.gen(images_a[0:4], images_b[0:4], noise.get('gen'))   # :238
            else:
                with torch.no_grad():
                    x_aa, x_ba, x_ab, x_bb, _ = self.gen(images_a[0:4], images_b[0:4], noise.get('gen'))
            f_aa, f_ba, f_ab, f_bb = self.dis.feats(x_aa, x_ba, x_ab, x_bb)
            zero = torch.zeros_like(f_aa)
            fl_a, fl_b = l1_mean(f_ab - f_aa, zero), l1_mean(f_ba - f_bb, zero)
            rl_a = reg('a', images_a, labels_a, noise.get('vae_a'))
            if mode == 4:
                rl_b = reg('b', images_b, labels_b, noise.get('vae_b'))
        total = hp['reg_w'] * (rl_a + rl_b) + hp['feature_w_reg'] * (fl_a + fl_b)
        total.backward()
        self.dis_opt.step()
        self.dis_reg_loss = (rl_a + rl_b).detach().numpy()
        self.dis_total_loss = total.detach().numpy()
        return (x_aa, x_ba, x_ab, x_bb, x_aa, x_bb, x_aa, x_bb)


# --------------------------------------------------------------------------------------
# joint read-out (depth_train.py:200-211,231-253 ; handpose_evaluation.py:97,130-136,203)
# --------------------------------------------------------------------------------------
NYU_EVAL_JOINTS = np.array([0, 3, 6, 9, 12, 15, 18, 21, 24, 25, 27, 30, 31, 32])   # depth_train.py:232


def joint_readout(dis, vae, images_b, labels_b, com, cube, nyu=True):
    """regress_b -> vae.decode -> mm joints; returns dict of joints, per-joint errors and metrics."""
    with torch.no_grad():
        _, post, _ = dis.regress_b(images_b)
        pose = vae.decode(post)
    n = labels_b.shape[0]
    gt = labels_b.detach().numpy().reshape(n, -1, 3)
    pr = pose.detach().numpy().reshape(n, -1, 3)
    if nyu:
        gt, pr = gt[:, NYU_EVAL_JOINTS], pr[:, NYU_EVAL_JOINTS]
    cube = np.asarray(cube, dtype=np.float32)
    com = np.asarray(com, dtype=np.float32).reshape(n, 1, 3)
    gt3d = gt * (cube[0] / 2.) + com
    pr3d = pr * (cube[0] / 2.) + com
    err = np.sqrt(np.square(gt3d - pr3d).sum(axis=2))            # [n, J]
    return dict(pose=pose.detach().numpy(), joints_mm=pr3d, err=err,
                mean_err=np.nanmean(np.nanmean(err, axis=1)),    # getMeanError, handpose_evaluation.py:97
                worst_joint=np.argmax(err, axis=1),              # behind getMaxErrorOverSeq, :130-136
                frames_within_40=int((np.nanmax(err, axis=1) <= 40).sum()))   # :203


# --------------------------------------------------------------------------------------
# harness hooks used by tests/golden/cases.py (NativeAdapter)
# --------------------------------------------------------------------------------------
def make_trainer(hp, device='cpu', literal=True):
    assert str(device) == 'cpu', "the oracle is CPU-only"
    return RefTrainer(hp, literal=literal)


def set_training(gen, flag):
    gen.training = bool(flag)


def named_grads(net, to_numpy):
    return OrderedDict((k, None if v.grad is None else to_numpy(v.grad)) for k, v in net.p.items())
