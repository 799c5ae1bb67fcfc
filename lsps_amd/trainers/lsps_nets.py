"""The four networks of the depth path on the HIP kernels (reference: src/trainers/lsps_nets.py).

Same class names, constructor dicts, method signatures and state-dict keys as the reference;
additional keyword-only `noise=` arguments expose the two random draws (GaussianNoiseLayer,
common_net.py:39; poseVAE.encode, lsps_nets.py:77) so parity tests can inject them.
"""
import torch
import torch.nn as nn

from .common_net import *  # noqa: F401,F403
from .common_net import (ACT_LRELU, ACT_NONE, ACT_TANH, Conv2d, ConvTranspose2d, GaussianNoiseLayer,
                         LeakyINSResBlock, LeakyINSResNeXtBlock, LeakyReLUConv2d, LeakyReLUConvTranspose2d,
                         LeakyReLULinear, Linear, _Fused, run_layers)
from .. import ops
from ..ops import ACT_SOFTPLUS


class _Net(nn.Module):
    """Base: `zero_grad` keeps the flat gradient arena (lsps_amd/optim.py) instead of dropping .grad;
    `.cuda(gpu)` moves the parameters like the reference's per-net overrides do."""

    _arena = None

    def zero_grad(self, set_to_none=True):
        if self._arena is not None:
            self._arena.zero_grad()
        else:
            super(_Net, self).zero_grad(set_to_none=set_to_none)

    def frozen(self):
        return _Frozen(self)


class _Frozen(object):
    """Context: parameters do not require grad (skips weight-gradient kernels nobody consumes)."""

    def __init__(self, net):
        self.params = [p for p in net.parameters() if p.requires_grad]

    def __enter__(self):
        for p in self.params:
            p.requires_grad_(False)

    def __exit__(self, *a):
        for p in self.params:
            p.requires_grad_(True)


class Mapping(_Net):
    """Pose code [n, input_dim] -> latent map [n, ch, dim, dim] by four 4x4 transposed convs (lsps_nets.py:8-31)."""

    def __init__(self, params):
        super(Mapping, self).__init__()
        self.input_dim = params['input_dim']
        dim, ch = params['output_dim'], params['output_ch']
        self.output_dim = (ch, dim, dim)
        self.model = nn.Sequential(
            LeakyReLUConvTranspose2d(self.input_dim, 4 * ch, kernel_size=4, stride=1, padding=0),
            LeakyReLUConvTranspose2d(4 * ch, 4 * ch, kernel_size=4, stride=2, padding=1),
            LeakyReLUConvTranspose2d(4 * ch, 2 * ch, kernel_size=4, stride=2, padding=1),
            ConvTranspose2d(2 * ch, ch, kernel_size=4, stride=2, padding=1))

    def forward(self, x):
        return self.model(x.unsqueeze(2).unsqueeze(3).contiguous())


class poseVAE(_Net):
    """Pose MLP: encode 108->50->(20,20) with reparameterisation, decode 20->50->108 (lsps_nets.py:34-83)."""

    def __init__(self, params):
        super(poseVAE, self).__init__()
        self.input_dim, self.z_dim, self.h_dim = params['input_dim'], params['z_dim'], params['h_dim']
        self.en_fc1 = Linear(self.input_dim, self.h_dim, act=ACT_LRELU)      # + self.lrelu, :74
        self.en_mu = Linear(self.h_dim, self.z_dim)
        self.en_sigma = Linear(self.h_dim, self.z_dim, act=ACT_SOFTPLUS)     # + self.softplus, :76
        self.de_fc1 = LeakyReLULinear(self.z_dim, self.h_dim)
        self.de_fc2 = Linear(self.h_dim, self.input_dim)
        self.preset_parameters()

    def preset_parameters(self):                                              # :55-59
        for m in (self.en_mu, self.en_sigma):
            m.weight.data.normal_(0, 0.002)
            m.bias.data.normal_(0, 0.002)

    def encode(self, y, noise=None):
        h = self.en_fc1(y)
        mu, sd = self.en_mu(h), self.en_sigma(h)
        if noise is None:                                                     # N(0, 0.05) ALWAYS, also in eval (:77)
            noise = torch.randn(mu.size(), device=mu.device, dtype=mu.dtype) * 0.05
        return mu + sd * noise, mu, sd

    def decode(self, z):
        if z.dim() == 1:                                                      # regress_*().squeeze() at n == 1
            return self.de_fc2(self.de_fc1(z.unsqueeze(0))).squeeze(0)
        return self.de_fc2(self.de_fc1(z))

    def forward(self, y, noise=None):
        z, mu, sd = self.encode(y, noise)
        return self.decode(z), z, mu, sd


class SharedDis(_Net):
    """Two per-domain strided-conv fronts, shared strided-conv trunk, D (1x1) and Post (2x2) heads
    (lsps_nets.py:86-160)."""

    def __init__(self, params):
        super(SharedDis, self).__init__()
        ch = params['ch']
        n_front, n_shared = params['n_front_layer'], params['n_shared_layer']
        n_expand = params['n_expand_layer'] if 'n_expand_layer' in params.keys() else 0
        self.post_dim, self.reg_dim = params['post_dim'], params['reg_dim']
        self.model_A, tch = self._make_front_net(ch, params['input_dim_a'], n_front)
        self.model_B, tch = self._make_front_net(ch, params['input_dim_b'], n_front)
        self.model_S, self.D, self.Post = self._make_shared_net(tch, n_shared, n_expand)
        self.dropout = None

    @staticmethod
    def _make_front_net(ch, input_dim, n_layer):
        layers = [LeakyReLUConv2d(input_dim, ch, kernel_size=7, stride=2, padding=3)]
        tch = ch
        for _ in range(1, n_layer):
            layers.append(LeakyReLUConv2d(tch, tch * 2, kernel_size=3, stride=2, padding=1))
            tch *= 2
        return nn.Sequential(*layers), tch

    def _make_shared_net(self, ch, n_layer, n_expand_layer):
        layers, tch = [], ch
        for _ in range(n_expand_layer):
            layers.append(LeakyReLUConv2d(tch, tch * 2, kernel_size=3, stride=1, padding=1))
            tch *= 2
        for _ in range(n_layer):
            layers.append(LeakyReLUConv2d(tch, tch * 2, kernel_size=3, stride=2, padding=1))
            tch *= 2
        post = Conv2d(tch, self.post_dim, kernel_size=2, stride=1, padding=0)
        discrim = Conv2d(tch, 1, kernel_size=1, stride=1, padding=0)
        return nn.Sequential(*layers), discrim, post

    def _trunk(self, f):
        """model_S on `f` ([N][C][H][W] f32, or a C8 tensor from the fronts in the bf16 math mode) -> f32 [N][C'][h][w].
        The trunk is a chain of 3x3 / stride-2 LeakyReLUConv2d on 16x16 ... 2x2 maps.  bf16 math mode: on bf16 tensors in the
        channel-group layout (csrc/c8s2.h, common_net.run_layers).  f32: where the geometry allows (N % 4 == 0, channels
        % 128 == 0, even maps) in batch-innermost layout on the plain-GEMM kernels of csrc/chwn.hip (one transpose in, one
        out); otherwise layer by layer in NCHW."""
        layers = list(self.model_S)
        if ops.is_c8(f) or ops.get_math_mode() == 'bf16':
            return ops.from_c8(run_layers(layers, f))
        N, C, H, W = ops._x3_shape(f) or f.shape             # f32 NCHW here (the fronts hand an X3 tensor only to a layer of their own chain)
        # three-limb operands on the bf16 matrix pipe (csrc/x3s2.h): the layers chain in the X3 layout (run_layers), the last one
        # hands out f32 NCHW; no transposes, the LeakyReLU backward passes ride in the dgrad epilogues.  EVERY trunk layer must
        # qualify (geometry and options.x3_min_gmac at its own map size): a chain that loses its tail to the plain NCHW kernels
        # pays a join / split per hand-over and never reaches the batch-innermost kernels below
        x3_all, c, h, w = bool(layers), C, H, W
        for l in layers:
            conv = l.model[0] if isinstance(l, LeakyReLUConv2d) else None
            x3_all = x3_all and conv is not None and ops.x3_conv_s2_ok(torch.empty((N, c, h, w), dtype=torch.float32, device='meta'),
                                                                       conv.weight, conv.stride, conv.padding)
            if not x3_all:
                break
            c, h, w = conv.weight.shape[0], h // 2, w // 2
        if x3_all:
            return ops.from_c8(run_layers(layers, f))
        # below ~96 samples the 128-wide batch tile of the GEMM is mostly padding (dis.feats on 16 samples: slower than NCHW)
        opt = ops.options.get()
        ok = len(layers) > 0 and N >= opt.chwn_min_n and opt.chwn
        c, h, w = C, H, W
        for l in layers:
            conv = l.model[0] if isinstance(l, LeakyReLUConv2d) else None
            ok = ok and conv is not None and tuple(conv.weight.shape[1:]) == (c, 3, 3) and conv.stride == 2 and \
                conv.padding == 1 and ops.conv3x3s2_chwn_ok(N, c, h, w, conv.weight.shape[0])
            if not ok:
                break
            c, h, w = conv.weight.shape[0], h // 2, w // 2
        if not ok:
            return self.model_S(f)
        t = ops.nchw_to_chwn(f)
        for l in layers:
            conv = l.model[0]
            t = ops.conv3x3s2_chwn(t, conv.weight, conv.bias, ops.ACT_LRELU, ops.LRELU_SLOPE)
        return ops.chwn_to_nchw(t)

    @staticmethod
    def _cat0(a, b):
        if ops.is_c8(a) != ops.is_c8(b):
            a, b = ops.from_c8(a), ops.from_c8(b)
        return torch.cat((a, b), 0)

    def _regress(self, front, x):
        post = self.Post(self._trunk(run_layers(front, x))).squeeze()
        return post, post, post

    def regress_a(self, x_A):
        return self._regress(self.model_A, x_A)

    def regress_b(self, x_B):
        return self._regress(self.model_B, x_B)

    def regress_feats(self, x_A, x_B, x_aa, x_ba, x_ab, x_bb):
        """`regress_a(x_A)[1]` (and `regress_b(x_B)[1]` when x_B is given) together with `feats(x_aa, x_ba, x_ab, x_bb)` from ONE
        pass of each front and ONE pass of the shared trunk: the samples of the two calls are concatenated along the batch
        (front A: [x_A | x_aa | x_ba], front B: [x_B | x_ab | x_bb]; every layer of the discriminator is per-sample, so the
        results are those of the separate calls up to the summation order of the weight gradients).  Round 5: the estimate
        modes call the two on 128 and 16 samples (lsps_trainer.py:238-247) and the 16-sample pass ran ~75 launches of kernels
        that fill a sixteenth of the chip; riding along in the 128-sample pass costs next to nothing.
        Returns (post_a, post_b or None, (f_aa, f_ba, f_ab, f_bb))."""
        na, nb = x_A.size(0), (x_B.size(0) if x_B is not None else 0)
        k = x_aa.size(0)
        fa = run_layers(self.model_A, torch.cat((x_A, x_aa, x_ba), 0))
        fb = run_layers(self.model_B, torch.cat((x_B, x_ab, x_bb), 0) if x_B is not None else torch.cat((x_ab, x_bb), 0))
        if ops.is_c8(fa) != ops.is_c8(fb):
            fa, fb = ops.from_c8(fa), ops.from_c8(fb)
        # batch order in the trunk: regression samples first (A, then B), then the four feature groups
        parts = (fa[:na], fb[:nb], fa[na:], fb[nb:])
        t = self._trunk(torch.cat([p for p in parts if p.size(0) > 0], 0))
        post = self.Post(t[:na + nb])
        post_a = post[:na].squeeze()
        post_b = post[na:].squeeze() if nb else None
        f = t[na + nb:]
        return post_a, post_b, torch.split(f, k, dim=0)

    def feats(self, x_aa, x_ba, x_ab, x_bb):
        f = self._cat0(run_layers(self.model_A, torch.cat((x_aa, x_ba), 0)), run_layers(self.model_B, torch.cat((x_ab, x_bb), 0)))
        f = self._trunk(f)
        return torch.split(f, f.size(0) // 4, dim=0)

    def forward(self, x_A, x_B, second_feats=False):
        f = self._trunk(self._cat0(run_layers(self.model_A, x_A), run_layers(self.model_B, x_B)))
        out_D = self.D(f)
        feats_A, feats_B = torch.split(f, f.size(0) // 2, dim=0)
        out_D_A, out_D_B = torch.split(out_D, out_D.size(0) // 2, dim=0)
        return out_D_A.reshape(-1), out_D_B.reshape(-1), feats_A, feats_B


class SharedResGen(_Net):
    """Per-domain encoders (7x7 stem, strided 3x3 downsamples, residual blocks) -> shared residual
    block + Gaussian noise -> shared latent -> shared residual block -> per-domain decoders
    (residual blocks, 3x3 stride-2 transposed convs, 1x1 transposed conv + tanh) (lsps_nets.py:164-272)."""

    def _res_block(self, tch, params):
        return LeakyINSResBlock(tch, tch, dropout=params.get('res_dropout_ratio', 0))   # lsps_nets.py:176-179

    def __init__(self, params):
        super(SharedResGen, self).__init__()
        ch = params['ch']
        res_block = lambda tch: self._res_block(tch, params)   # noqa: E731

        def encoder(input_dim):
            layers = [LeakyReLUConv2d(input_dim, ch, kernel_size=7, stride=1, padding=3)]
            tch = ch
            for _ in range(1, params['n_enc_front_blk']):
                layers.append(LeakyReLUConv2d(tch, tch * 2, kernel_size=3, stride=2, padding=1))
                tch *= 2
            layers += [res_block(tch) for _ in range(params['n_enc_res_blk'])]
            return nn.Sequential(*layers), tch

        def decoder(tch, output_dim):
            layers = [res_block(tch) for _ in range(params['n_gen_res_blk'])]
            for _ in range(1, params['n_gen_front_blk']):
                layers.append(LeakyReLUConvTranspose2d(tch, tch // 2, kernel_size=3, stride=2, padding=1,
                                                       output_padding=1))
                tch //= 2
            layers += [ConvTranspose2d(tch, output_dim, kernel_size=1, stride=1, padding=0, act=ACT_TANH),
                       _Fused('Tanh')]
            return nn.Sequential(*layers)

        self.encode_A, tch = encoder(params['input_dim_a'])
        self.encode_B, tch = encoder(params['input_dim_b'])
        self.enc_shared = nn.Sequential(*([res_block(tch) for _ in range(params['n_enc_shared_blk'])]
                                          + [GaussianNoiseLayer()]))
        self.dec_shared = nn.Sequential(*[res_block(tch) for _ in range(params['n_gen_shared_blk'])])
        self.decode_A = decoder(tch, params['input_dim_a'])
        self.decode_B = decoder(tch, params['input_dim_b'])

    # In the bf16 math mode the residual trunk (encoder blocks -> shared blocks + noise -> decoder blocks) runs on bf16
    # tensors in the channel-group layout of csrc/c8conv.h (common_net.run_layers); what the methods return is f32 NCHW.
    def _enc_shared(self, h, noise):
        return run_layers(self.enc_shared, h, noise)

    def decode(self, z):
        out = run_layers(self.dec_shared, z)
        return ops.from_c8(run_layers(self.decode_A, out)), ops.from_c8(run_layers(self.decode_B, out))

    def encode(self, x_A, x_B, noise_a=None, noise_b=None):
        return (ops.from_c8(self._enc_shared(run_layers(self.encode_A, x_A), noise_a)),
                ops.from_c8(self._enc_shared(run_layers(self.encode_B, x_B), noise_b)))

    def encode_pre(self, x_A, x_B):
        """The deterministic part of `forward`'s encoder half: both encoders and the residual blocks of `enc_shared`, i.e. the
        input of its GaussianNoiseLayer.  `forward(..., pre=...)` continues from it.  (LSPSTrainer's opt-in encoder sharing:
        dis_update and gen_update of one iteration call `gen(images_a, images_b)` with the same images and weights and differ only
        in the noise draw, lsps_trainer.py:86,145.)"""
        ha, hb = run_layers(self.encode_A, x_A), run_layers(self.encode_B, x_B)
        if ops.is_c8(ha) != ops.is_c8(hb):
            ha, hb = ops.from_c8(ha), ops.from_c8(hb)
        return run_layers([l for l in self.enc_shared if not isinstance(l, GaussianNoiseLayer)], torch.cat((ha, hb), 0))

    def forward(self, x_A, x_B, noise=None, pre=None):
        if pre is not None:
            shared = run_layers([l for l in self.enc_shared if isinstance(l, GaussianNoiseLayer)], pre, noise)
        else:
            ha, hb = run_layers(self.encode_A, x_A), run_layers(self.encode_B, x_B)
            if ops.is_c8(ha) != ops.is_c8(hb):
                ha, hb = ops.from_c8(ha), ops.from_c8(hb)
            out = torch.cat((ha, hb), 0)
            shared = self._enc_shared(out, noise)
        out = run_layers(self.dec_shared, shared)
        out_A, out_B = ops.from_c8(run_layers(self.decode_A, out)), ops.from_c8(run_layers(self.decode_B, out))
        x_Aa, x_Ba = torch.split(out_A, x_A.size(0), dim=0)
        x_Ab, x_Bb = torch.split(out_B, x_A.size(0), dim=0)
        return x_Aa, x_Ba, x_Ab, x_Bb, ops.from_c8(shared)

    def forward_a2b(self, x_A, noise=None):
        shared = self._enc_shared(run_layers(self.encode_A, x_A), noise)
        return ops.from_c8(run_layers(self.decode_B, run_layers(self.dec_shared, shared))), ops.from_c8(shared)

    def forward_b2a(self, x_B, noise=None):
        shared = self._enc_shared(run_layers(self.encode_B, x_B), noise)
        return ops.from_c8(run_layers(self.decode_A, run_layers(self.dec_shared, shared))), ops.from_c8(shared)


class SharedResXGen(SharedResGen):
    """SharedResGen with ResNeXt residual blocks (1x1 expand, grouped 3x3, 1x1 project; lsps_nets.py:277-387).
    Same topology, methods and state-dict keys as the reference class; not used by the shipped configs."""

    def _res_block(self, tch, params):
        k = params['n_resnext_k'] if 'n_resnext_k' in params.keys() else 1
        c = params['n_resnext_c'] if 'n_resnext_c' in params.keys() else 4
        return LeakyINSResNeXtBlock(tch, tch, k=k, cardinality=c, dropout=params.get('res_dropout_ratio', 0))
