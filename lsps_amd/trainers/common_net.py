"""Building blocks of the depth path on the HIP kernels (reference: src/trainers/common_net.py).

Class names, constructor signatures and state-dict keys follow the reference so that its
checkpoints load unchanged (`<block>.model.0.weight`, `.model.3.weight` ...), but every forward
runs fused liblsps_hip kernels: conv + bias + LeakyReLU in one launch, InstanceNorm + LeakyReLU
(+ residual add) in one in-place pass.  Parameter-free placeholder modules keep the Sequential
indices of the reference where an op has been fused into its neighbour.
"""
import math

import numpy as np  # noqa: F401  (leaked by the reference's namespace; depth_train.py:220 uses `np`)
import torch
import torch.nn as nn
from torch.autograd import Variable  # noqa: F401  (leaked too; depth_train.py:145 uses it)

from .init import *  # noqa: F401,F403
from .init import gaussian_weights_init
from .. import ops
from ..ops import ACT_LRELU, ACT_NONE, ACT_TANH, LRELU_SLOPE  # noqa: F401


class _Fused(nn.Module):
    """Index-keeping placeholder for an op executed inside the neighbouring fused kernel."""

    def __init__(self, what):
        super(_Fused, self).__init__()
        self.what = what

    def forward(self, x):
        return x

    def extra_repr(self):
        return 'fused: %s' % self.what


def _mark_dead_bias(conv):
    """The bias of a conv that feeds an affine-free InstanceNorm cancels exactly and is not applied here (its gradient is
    exactly 0).  The reference still passes it through Adam — round-off-sized gradient plus the coupled weight decay — so the
    trainer lets the optimizer see it (zero gradient + weight decay) whenever the conv's weight got a gradient."""
    if getattr(conv, 'bias', None) is not None:
        conv.bias._lsps_dead_of = conv.weight


def _default_reset(weight, bias, fan_in):
    nn.init.kaiming_uniform_(weight, a=math.sqrt(5))          # torch's default conv / linear init
    if bias is not None:
        bound = 1.0 / math.sqrt(fan_in) if fan_in > 0 else 0.0
        nn.init.uniform_(bias, -bound, bound)


class Conv2d(nn.Module):
    """Parameter holder + launcher for lsps_conv2d_* (replaces nn.Conv2d; weight (K,C,R,S))."""

    def __init__(self, n_in, n_out, kernel_size, stride=1, padding=0, bias=True, act=ACT_NONE):
        super(Conv2d, self).__init__()
        self.stride, self.padding, self.act = stride, padding, act
        self.weight = nn.Parameter(torch.empty(n_out, n_in, kernel_size, kernel_size))
        self.bias = nn.Parameter(torch.empty(n_out)) if bias else None
        _default_reset(self.weight, self.bias, n_in * kernel_size * kernel_size)

    def forward(self, x):
        return ops.conv2d(x, self.weight, self.bias, self.stride, self.padding, self.act, LRELU_SLOPE)


class ConvTranspose2d(nn.Module):
    """Parameter holder + launcher for lsps_convT2d_* (replaces nn.ConvTranspose2d; weight (Ci,Co,R,S))."""

    def __init__(self, n_in, n_out, kernel_size, stride=1, padding=0, output_padding=0, bias=True, act=ACT_NONE):
        super(ConvTranspose2d, self).__init__()
        self.stride, self.padding, self.output_padding, self.act = stride, padding, output_padding, act
        self.weight = nn.Parameter(torch.empty(n_in, n_out, kernel_size, kernel_size))
        self.bias = nn.Parameter(torch.empty(n_out)) if bias else None
        _default_reset(self.weight, self.bias, n_out * kernel_size * kernel_size)

    def forward(self, x):
        return ops.conv_transpose2d(x, self.weight, self.bias, self.stride, self.padding, self.output_padding,
                                    self.act, LRELU_SLOPE)


class Linear(nn.Module):
    """Parameter holder + launcher for lsps_linear_* (replaces nn.Linear)."""

    def __init__(self, n_in, n_out, act=ACT_NONE):
        super(Linear, self).__init__()
        self.act = act
        self.weight = nn.Parameter(torch.empty(n_out, n_in))
        self.bias = nn.Parameter(torch.empty(n_out))
        _default_reset(self.weight, self.bias, n_in)

    def forward(self, x):
        return ops.linear(x, self.weight, self.bias, self.act, LRELU_SLOPE)


class GaussianNoiseLayer(nn.Module):
    """x + N(0,1) in training, identity in eval (common_net.py:32-40).  `noise` injects the draw."""

    def forward(self, x, noise=None):
        if not self.training:
            return x
        if ops.is_c8(x):                                  # bf16 trunk: the draw has x's (N, C, H, W)
            N, G, H, W, _ = x.shape
            if noise is None:
                noise = torch.randn((N, G * 8, H, W), device=x.device, dtype=torch.float32)
            return ops.add_c8(x, noise.detach())           # the f32 draw is added as it is (no conversion pass)
        if noise is None:
            noise = torch.randn(x.size(), device=x.device, dtype=x.dtype)
        return ops.axpy(x, noise, 1.0)


def run_layers(layers, x, noise=None):
    """Applies a sequence of layers.  In the bf16 math mode the layers that have kernels for bf16 tensors in the
    channel-group layout of csrc/c8conv.h / c8s2.h — LeakyINSResBlock chains on 32x32 maps, the 3x3 / stride-2
    LeakyReLUConv2d and LeakyReLUConvTranspose2d layers, GaussianNoiseLayer — run on such tensors: `x` is converted in front
    of the first of them, stays in that layout through consecutive ones and is converted back in front of any other layer
    (the caller converts what it hands out: ops.from_c8)."""
    prev = None            # ops.ActHolder of the C8 / X3 layer whose output `x` is (consecutive layers: nobody else consumes it)
    layers = list(layers)

    def x3_next(i, y_shape):
        """Does layer i + 1 take the X3 output (shape as f32 NCHW: `y_shape`) of layer i?  Then layer i writes X3 and the pair
        shares an ActHolder (layer i + 1's dgrad applies layer i's LeakyReLU backward)."""
        if i + 1 >= len(layers) or not ops.x3_fuse_enabled():
            return False
        n = layers[i + 1]
        probe = torch.empty(y_shape, dtype=torch.float32, device='meta')
        if isinstance(n, LeakyReLUConv2d):
            c = n.model[0]
            return ops.x3_conv_s2_ok(probe, c.weight, c.stride, c.padding)
        if isinstance(n, LeakyReLUConvTranspose2d) and isinstance(layers[i], LeakyReLUConvTranspose2d):
            c = n.model[0]
            return c.act == ACT_LRELU and ops.x3_convT_s2_ok(probe, c.weight, c.stride, c.padding, c.output_padding)
        return False

    for i, l in enumerate(layers):
        if isinstance(l, LeakyINSResBlock):
            ch = l.model[0].weight.shape[0]
            drop = l.dropout if (l.training and l.dropout > 0) else 0.0
            if l.model[0].stride == 1 and l.model[0].weight.shape[1] == ch and ops.c8_block_ok(x, ch, drop):
                x = ops.to_c8(x)
            else:
                x = ops.from_c8(x)
            x, prev = l(x), None
        elif isinstance(l, LeakyReLUConv2d):
            c = l.model[0]
            if ops.c8_stem_ok(x, c.weight, c.stride, c.padding):     # 7x7 stem: f32 image in, C8 activation out
                x, prev = ops.stem_c8(x, c.weight, c.bias, c.stride, c.padding, LRELU_SLOPE), None
            elif ops.c8_conv_s2_ok(x, c.weight, c.stride, c.padding):
                xin = ops.to_c8(x)
                own = ops.ActHolder(LRELU_SLOPE)
                x = ops.conv3x3s2_c8(xin, c.weight, c.bias, LRELU_SLOPE, prev if xin is x else None, own)
                prev = own
            elif ops.x3_stem_ok(x, c.weight, c.stride, c.padding) and \
                    x3_next(i, (x.shape[0], c.weight.shape[0], ops.conv_out_size(x.shape[2], c.weight.shape[2], c.stride, c.padding),
                                ops.conv_out_size(x.shape[3], c.weight.shape[3], c.stride, c.padding))):
                # f32 math mode, 7x7 stem in front of a three-limb layer: the f32 kernel writes its activation as limbs; that layer's
                # dgrad applies this one's LeakyReLU backward and hands the f32 gradient back through the holder (ops.ActHolder)
                own = ops.ActHolder(LRELU_SLOPE, want='f32')
                x = ops.stem_x3(x, c.weight, c.bias, c.stride, c.padding, LRELU_SLOPE, own)
                prev = own
            elif ops.x3_conv_s2_ok(x, c.weight, c.stride, c.padding):   # f32 math mode: three-limb operands on the bf16 pipe
                N, C, H, W = ops._x3_shape(x)
                chain = x3_next(i, (N, c.weight.shape[0], H // 2, W // 2))
                own = ops.ActHolder(LRELU_SLOPE) if chain else None
                x = ops.conv3x3s2_x3(x, c.weight, c.bias, LRELU_SLOPE, prev if ops.is_x3(x) else None, own, out_f32=not chain)
                prev = own
            else:
                x, prev = l(ops.from_c8(x)), None
        elif isinstance(l, LeakyReLUConvTranspose2d):
            c = l.model[0]
            if ops.c8_convT_s2_ok(x, c.weight, c.stride, c.padding, c.output_padding):
                xin = ops.to_c8(x)
                own = ops.ActHolder(LRELU_SLOPE)
                x = ops.convT3x3s2_c8(xin, c.weight, c.bias, LRELU_SLOPE, prev if xin is x else None, own)
                prev = own
            elif c.act == ACT_LRELU and ops.x3_convT_s2_ok(x, c.weight, c.stride, c.padding, c.output_padding):
                # X3 out into another three-limb transposed conv (its dgrad applies this layer's LeakyReLU backward); f32 out into
                # the 1x1 head, which fuses this layer's LeakyReLU backward like on the f32 path below and hands the gradient back
                # as limbs (ActHolder.want)
                N, C, H, W = ops._x3_shape(x)
                chain = x3_next(i, (N, c.weight.shape[1], 2 * H, 2 * W))
                own = ops.ActHolder(LRELU_SLOPE, want=None if chain else 'x3')
                x = ops.convT3x3s2_x3(x, c.weight, c.bias, LRELU_SLOPE, prev if ops.is_x3(x) else None, own, out_f32=not chain)
                prev = own
            else:                                                       # f32 (or a shape without a C8 kernel)
                own = ops.ActHolder(LRELU_SLOPE)
                x = ops.conv_transpose2d(ops.from_c8(x), c.weight, c.bias, c.stride, c.padding, c.output_padding, c.act, LRELU_SLOPE,
                                         None, own if c.act == ACT_LRELU else None)
                prev = own if c.act == ACT_LRELU else None
        elif isinstance(l, GaussianNoiseLayer):
            x, prev = l(x, noise), None
        elif isinstance(l, ConvTranspose2d) and ops.c8_pw1_ok(x, l.weight, l.stride, l.padding, l.output_padding):
            x, prev = ops.pw1_c8(x, l.weight, l.bias, l.act, LRELU_SLOPE, prev), None      # 1x1 output head (+ Tanh) straight from C8
        elif isinstance(l, ConvTranspose2d):                         # f32 output head: the fused form needs the 1x1 / one-channel geometry
            xin = ops.from_c8(x)
            x = ops.conv_transpose2d(xin, l.weight, l.bias, l.stride, l.padding, l.output_padding, l.act, LRELU_SLOPE,
                                     prev if xin is x else None, None)
            prev = None
        elif isinstance(l, _Fused):
            pass
        else:
            x, prev = l(ops.from_c8(x)), None
    return x


class LeakyINSResBlock(nn.Module):
    """x + IN(conv3x3(LReLU(IN(conv3x3(x))))) (common_net.py:160-181).
    The conv biases are kept for state-dict parity but not applied: an affine-free InstanceNorm
    subtracts the per-plane mean, so a per-channel constant cancels exactly (gradient exactly 0)."""

    def __init__(self, inplanes, planes, stride=1, dropout=0.0):
        super(LeakyINSResBlock, self).__init__()
        layers = [Conv2d(inplanes, planes, 3, stride, 1), _Fused('InstanceNorm2d'), _Fused('LeakyReLU'),
                  Conv2d(planes, planes, 3, 1, 1), _Fused('InstanceNorm2d + residual add')]
        self.dropout = float(dropout)
        if dropout > 0:
            layers.append(_Fused('Dropout'))            # same child count as the reference (common_net.py:171-172)
        self.model = nn.Sequential(*layers)
        self.model.apply(gaussian_weights_init)
        _mark_dead_bias(self.model[0])
        _mark_dead_bias(self.model[3])

    def forward(self, x, drop_mask=None):
        """`drop_mask` (tests): the keep mask ALREADY divided by 1-p; default: drawn here in training mode."""
        c1, c2 = self.model[0], self.model[3]
        dropping = self.dropout > 0 and (self.training or drop_mask is not None)
        if ops.is_c8(x):
            # bf16 trunk (math mode 'bf16'): activations stay in the channel-group layout between the blocks (run_layers)
            assert not dropping and c1.stride == 1, "the C8 residual path has no dropout / stride"
            return ops.res_block_c8(x, c1.weight, c2.weight)
        if not dropping and c1.stride == 1 and c1.weight.shape[0] == c1.weight.shape[1]:
            # one autograd node (fused conv + InstanceNorm entries, skip gradient added in the dgrad epilogue); also under
            # no_grad (dis_update's generator pass, evaluation): nothing is saved then, the fused forward is the same
            return ops.res_block(x, c1.weight, c2.weight)
        h = ops.conv2d(x, c1.weight, None, c1.stride, 1)
        h = ops.instance_norm_(h, None, LRELU_SLOPE)
        h = ops.conv2d(h, c2.weight, None, 1, 1)
        if dropping:
            h = ops.instance_norm_(h, None, -1.0)
            if drop_mask is None:
                drop_mask = (torch.rand_like(h) >= self.dropout).to(h.dtype) / (1.0 - self.dropout)
            return ops.mul_add(x, h, drop_mask)
        return ops.instance_norm_(h, x, -1.0)


class LeakyReLUConv2d(nn.Module):
    """LeakyReLU(conv(x)) in one launch (common_net.py:246-256)."""

    def __init__(self, n_in, n_out, kernel_size, stride, padding=0):
        super(LeakyReLUConv2d, self).__init__()
        self.model = nn.Sequential(Conv2d(n_in, n_out, kernel_size, stride, padding, act=ACT_LRELU),
                                   _Fused('LeakyReLU'))
        self.model.apply(gaussian_weights_init)

    def forward(self, x):
        c = self.model[0]
        if ops.is_c8(x):                                  # bf16 math mode, 3x3 / stride 2 (run_layers decides): csrc/c8s2.h
            return ops.conv3x3s2_c8(x, c.weight, c.bias, LRELU_SLOPE)
        return c(x)


class LeakyReLUConvTranspose2d(nn.Module):
    """LeakyReLU(conv_transpose(x)) in one launch per output parity class (common_net.py:258-268)."""

    def __init__(self, n_in, n_out, kernel_size, stride, padding=0, output_padding=0):
        super(LeakyReLUConvTranspose2d, self).__init__()
        self.model = nn.Sequential(
            ConvTranspose2d(n_in, n_out, kernel_size, stride, padding, output_padding, act=ACT_LRELU),
            _Fused('LeakyReLU'))
        self.model.apply(gaussian_weights_init)

    def forward(self, x):
        c = self.model[0]
        if ops.is_c8(x):
            return ops.convT3x3s2_c8(x, c.weight, c.bias, LRELU_SLOPE)
        return c(x)


class LeakyReLULinear(nn.Module):
    """LeakyReLU(linear(x)) (common_net.py:221-231)."""

    def __init__(self, n_in, n_out):
        super(LeakyReLULinear, self).__init__()
        self.model = nn.Sequential(Linear(n_in, n_out, act=ACT_LRELU), _Fused('LeakyReLU'))

    def forward(self, x):
        return self.model[0](x)


# ---------------------------------------------------------------------------------------------
# Variants that no shipped config instantiates (reference: common_net.py:42-135,183-379).  The ones that
# can be expressed with the kernels of the hot path are thin compositions of them; BatchNorm- / ReLU- /
# cv2-based ones keep their names in the namespace (the drop-in contract) but refuse
# construction: there is no HIP kernel behind them and no silent torch fallback.
# ---------------------------------------------------------------------------------------------
class Conv2dGrouped(nn.Module):
    """Grouped 3x3 conv of the ResNeXt block (common_net.py:116): weight (n_out, n_in/groups, k, k).
    ops.conv2d_grouped: one launch per group on the channel slices of the full tensors (lsps_conv2d_grouped_*)."""

    def __init__(self, n_in, n_out, kernel_size, stride=1, padding=0, groups=1, bias=True):
        super(Conv2dGrouped, self).__init__()
        assert n_in % groups == 0 and n_out % groups == 0
        self.stride, self.padding, self.groups = stride, padding, groups
        self.weight = nn.Parameter(torch.empty(n_out, n_in // groups, kernel_size, kernel_size))
        self.bias = nn.Parameter(torch.empty(n_out)) if bias else None
        _default_reset(self.weight, self.bias, (n_in // groups) * kernel_size * kernel_size)

    def forward(self, x, use_bias=True):
        b = self.bias if (use_bias and self.bias is not None) else None
        return ops.conv2d_grouped(x, self.weight, b, self.stride, self.padding, self.groups)


class LeakyINSResNeXtBlock(nn.Module):
    """x + IN(conv1x1(LReLU(IN(gconv3x3(LReLU(IN(conv1x1(x)))))))) (common_net.py:111-132)."""

    def __init__(self, inplanes, planes, k=2, cardinality=8, dropout=0.0):
        super(LeakyINSResNeXtBlock, self).__init__()
        layers = [Conv2d(inplanes, k * inplanes, 1, 1, 0), _Fused('InstanceNorm2d'), _Fused('LeakyReLU'),
                  Conv2dGrouped(k * inplanes, k * inplanes, 3, 1, 1, groups=cardinality), _Fused('InstanceNorm2d'),
                  _Fused('LeakyReLU'), Conv2d(k * inplanes, planes, 1, 1, 0), _Fused('InstanceNorm2d + residual add')]
        self.dropout = float(dropout)
        if dropout > 0:
            layers.append(_Fused('Dropout'))            # common_net.py:123-124
        self.model = nn.Sequential(*layers)
        self.model.apply(gaussian_weights_init)

    def forward(self, x, drop_mask=None):
        c1, c2, c3 = self.model[0], self.model[3], self.model[6]
        h = ops.instance_norm_(ops.conv2d(x, c1.weight, None, 1, 0), None, LRELU_SLOPE)   # biases cancel under IN
        h = ops.instance_norm_(c2(h, use_bias=False), None, LRELU_SLOPE)
        h = ops.conv2d(h, c3.weight, None, 1, 0)
        if self.dropout > 0 and (self.training or drop_mask is not None):
            h = ops.instance_norm_(h, None, -1.0)
            if drop_mask is None:
                drop_mask = (torch.rand_like(h) >= self.dropout).to(h.dtype) / (1.0 - self.dropout)
            return ops.mul_add(x, h, drop_mask)
        return ops.instance_norm_(h, x, -1.0)


class LeakyReLUINSConv2d(nn.Module):
    """LReLU(IN(conv(x))) (common_net.py:324-335)."""

    def __init__(self, n_in, n_out, kernel_size, stride, padding=0):
        super(LeakyReLUINSConv2d, self).__init__()
        self.model = nn.Sequential(Conv2d(n_in, n_out, kernel_size, stride, padding), _Fused('InstanceNorm2d'),
                                   _Fused('LeakyReLU'))
        self.model.apply(gaussian_weights_init)

    def forward(self, x):
        c = self.model[0]
        return ops.instance_norm_(ops.conv2d(x, c.weight, None, c.stride, c.padding), None, LRELU_SLOPE)


class LeakyReLUINSConvTranspose2d(nn.Module):
    """LReLU(IN(conv_transpose(x))) (common_net.py:337-349)."""

    def __init__(self, n_in, n_out, kernel_size, stride, padding, output_padding):
        super(LeakyReLUINSConvTranspose2d, self).__init__()
        self.model = nn.Sequential(ConvTranspose2d(n_in, n_out, kernel_size, stride, padding, output_padding),
                                   _Fused('InstanceNorm2d'), _Fused('LeakyReLU'))
        self.model.apply(gaussian_weights_init)

    def forward(self, x):
        c = self.model[0]
        y = ops.conv_transpose2d(x, c.weight, None, c.stride, c.padding, c.output_padding)
        return ops.instance_norm_(y, None, LRELU_SLOPE)


class LeakyReLUResBlock(nn.Module):
    """x + conv(LReLU(conv(x))) (common_net.py:201-215; both convs map n_in -> n_out as in the reference)."""

    def __init__(self, n_in, n_out, kernel_size, stride, padding=0):
        super(LeakyReLUResBlock, self).__init__()
        self.model = nn.Sequential(Conv2d(n_in, n_out, kernel_size, stride, padding, act=ACT_LRELU), _Fused('LeakyReLU'),
                                   Conv2d(n_in, n_out, kernel_size, stride, padding))
        self.model.apply(gaussian_weights_init)

    def forward(self, x):
        return ops.axpy(self.model[2](self.model[0](x)), x, 1.0)


class Bias2d(nn.Module):
    """Per-channel bias (common_net.py:92-103)."""

    def __init__(self, channels):
        super(Bias2d, self).__init__()
        self.bias = nn.Parameter(torch.empty(channels).normal_(0, 0.002))

    def forward(self, x):
        return x + self.bias.view(1, -1, 1, 1)


class GaussianVAE(nn.Module):
    """mu / softplus(sd) heads (common_net.py:42-65)."""

    def __init__(self, n_in, n_out):
        super(GaussianVAE, self).__init__()
        self.en_mu, self.en_sigma = Linear(n_in, n_out), Linear(n_in, n_out, act=ops.ACT_SOFTPLUS)
        for m in (self.en_mu, self.en_sigma):
            m.weight.data.normal_(0, 0.002)
            m.bias.data.normal_(0, 0.002)

    def forward(self, x):
        return self.en_mu(x), self.en_sigma(x)

    def sample(self, x, noise=None):
        mu, sd = self.forward(x)
        if noise is None:
            noise = torch.randn(mu.size(), device=mu.device, dtype=mu.dtype)
        return mu + sd * noise, mu, sd


class _BatchNorm(nn.Module):
    """State holder for lsps_bnorm_* with torch's BatchNorm state-dict keys (weight, bias, running_mean, running_var,
    num_batches_tracked); eps 1e-5, momentum 0.1 (the reference uses the defaults everywhere)."""

    def __init__(self, num_features, affine=True):
        super(_BatchNorm, self).__init__()
        self.num_features, self.affine = num_features, affine
        if affine:
            self.weight = nn.Parameter(torch.ones(num_features))
            self.bias = nn.Parameter(torch.zeros(num_features))
        else:
            self.register_parameter('weight', None)
            self.register_parameter('bias', None)
        self.register_buffer('running_mean', torch.zeros(num_features))
        self.register_buffer('running_var', torch.ones(num_features))
        self.register_buffer('num_batches_tracked', torch.tensor(0, dtype=torch.long))

    def forward(self, x, slope=-1.0, extra_bias=None):
        """act(BN(x) [+ extra_bias]): `extra_bias` is the Bias2d that follows an affine-free BN in the "BNNS" blocks."""
        if self.training:
            self.num_batches_tracked += 1
        beta = self.bias if self.affine else extra_bias
        return ops.batch_norm(x, self.weight, beta, self.running_mean, self.running_var, self.training, slope)


class BatchNorm2d(_BatchNorm):
    pass


class BatchNorm1d(_BatchNorm):
    pass


class LeakyReLUBNConv2d(nn.Module):
    """LReLU(BN(conv(x))), conv without bias (common_net.py:270-281)."""

    def __init__(self, n_in, n_out, kernel_size, stride, padding=0):
        super(LeakyReLUBNConv2d, self).__init__()
        self.model = nn.Sequential(Conv2d(n_in, n_out, kernel_size, stride, padding, bias=False), BatchNorm2d(n_out),
                                   _Fused('LeakyReLU'))
        self.model.apply(gaussian_weights_init)

    def forward(self, x):
        return self.model[1](self.model[0](x), LRELU_SLOPE)


class LeakyReLUBNConvTranspose2d(nn.Module):
    """LReLU(BN(conv_transpose(x))) (common_net.py:283-294)."""

    def __init__(self, n_in, n_out, kernel_size, stride, padding=0, output_padding=0):
        super(LeakyReLUBNConvTranspose2d, self).__init__()
        self.model = nn.Sequential(ConvTranspose2d(n_in, n_out, kernel_size, stride, padding, output_padding, bias=False),
                                   BatchNorm2d(n_out), _Fused('LeakyReLU'))
        self.model.apply(gaussian_weights_init)

    def forward(self, x):
        return self.model[1](self.model[0](x), LRELU_SLOPE)


class LeakyReLUBNNSConv2d(nn.Module):
    """LReLU(BN_noaffine(conv(x)) + Bias2d) (common_net.py:296-308): the bias rides in the BN kernel as beta."""

    def __init__(self, n_in, n_out, kernel_size, stride, padding=0):
        super(LeakyReLUBNNSConv2d, self).__init__()
        self.model = nn.Sequential(Conv2d(n_in, n_out, kernel_size, stride, padding, bias=True),
                                   BatchNorm2d(n_out, affine=False), Bias2d(n_out), _Fused('LeakyReLU'))
        self.model.apply(gaussian_weights_init)

    def forward(self, x):
        return self.model[1](self.model[0](x), LRELU_SLOPE, self.model[2].bias)


class LeakyReLUBNNSConvTranspose2d(nn.Module):
    """LReLU(BN_noaffine(conv_transpose(x)) + Bias2d) (common_net.py:310-322)."""

    def __init__(self, n_in, n_out, kernel_size, stride, padding=0):
        super(LeakyReLUBNNSConvTranspose2d, self).__init__()
        self.model = nn.Sequential(ConvTranspose2d(n_in, n_out, kernel_size, stride, padding, bias=True),
                                   BatchNorm2d(n_out, affine=False), Bias2d(n_out), _Fused('LeakyReLU'))
        self.model.apply(gaussian_weights_init)

    def forward(self, x):
        return self.model[1](self.model[0](x), LRELU_SLOPE, self.model[2].bias)


class LeakyReLUBNLinear(nn.Module):
    """LReLU(BN1d_noaffine(linear(x))) (common_net.py:233-244)."""

    def __init__(self, n_in, n_out):
        super(LeakyReLUBNLinear, self).__init__()
        self.model = nn.Sequential(Linear(n_in, n_out), BatchNorm1d(n_out, affine=False), _Fused('LeakyReLU'))
        self.model.apply(gaussian_weights_init)

    def forward(self, x):
        return self.model[1](self.model[0](x), LRELU_SLOPE)


class LeakyReLUBNNSResBlock(nn.Module):
    """x + BN(conv(LReLU(BN(conv(x))))), affine-free BN, convs without bias (common_net.py:183-199)."""

    def __init__(self, n_in, n_out, kernel_size, stride, padding=0):
        super(LeakyReLUBNNSResBlock, self).__init__()
        self.model = nn.Sequential(Conv2d(n_in, n_out, kernel_size, stride, padding, bias=False),
                                   BatchNorm2d(n_out, affine=False), _Fused('LeakyReLU'),
                                   Conv2d(n_in, n_out, kernel_size, stride, padding, bias=False),
                                   BatchNorm2d(n_out, affine=False))
        self.model.apply(gaussian_weights_init)

    def forward(self, x):
        h = self.model[1](self.model[0](x), LRELU_SLOPE)
        return ops.axpy(self.model[4](self.model[3](h)), x, 1.0)


class INSResBlock(nn.Module):
    """x + [Dropout](IN(conv3x3(ReLU(IN(conv3x3(x)))))) (common_net.py:137-158).  ReLU zeroes half of the normalised
    values, so the InstanceNorm backward cannot be computed from the activation's output as in the LeakyReLU block:
    the normalised tensor is kept and the ReLU is a separate pass."""

    def __init__(self, inplanes, planes, stride=1, dropout=0.0):
        super(INSResBlock, self).__init__()
        layers = [Conv2d(inplanes, planes, 3, stride, 1), _Fused('InstanceNorm2d'), _Fused('ReLU'),
                  Conv2d(planes, planes, 3, 1, 1), _Fused('InstanceNorm2d + residual add')]
        self.dropout = float(dropout)
        if dropout > 0:
            layers.append(_Fused('Dropout'))
        self.model = nn.Sequential(*layers)
        self.model.apply(gaussian_weights_init)

    def forward(self, x, drop_mask=None):
        c1, c2 = self.model[0], self.model[3]
        h = ops.instance_norm_(ops.conv2d(x, c1.weight, None, c1.stride, 1), None, -1.0)     # biases cancel under IN
        h = ops.conv2d(ops.act(h, ACT_LRELU, 0.0), c2.weight, None, 1, 1)
        if self.dropout > 0 and (self.training or drop_mask is not None):
            h = ops.instance_norm_(h, None, -1.0)
            if drop_mask is None:
                drop_mask = (torch.rand_like(h) >= self.dropout).to(h.dtype) / (1.0 - self.dropout)
            return ops.mul_add(x, h, drop_mask)
        return ops.instance_norm_(h, x, -1.0)


class ReLUINSConv2d(nn.Module):
    """ReLU(IN(conv(x))) (common_net.py:354-365)."""

    def __init__(self, n_in, n_out, kernel_size, stride, padding=0):
        super(ReLUINSConv2d, self).__init__()
        self.model = nn.Sequential(Conv2d(n_in, n_out, kernel_size, stride, padding), _Fused('InstanceNorm2d'),
                                   _Fused('ReLU'))
        self.model.apply(gaussian_weights_init)

    def forward(self, x):
        c = self.model[0]
        return ops.act(ops.instance_norm_(ops.conv2d(x, c.weight, None, c.stride, c.padding), None, -1.0), ACT_LRELU, 0.0)


class ReLUINSConvTranspose2d(nn.Module):
    """ReLU(IN(conv_transpose(x))) (common_net.py:367-380)."""

    def __init__(self, n_in, n_out, kernel_size, stride, padding, output_padding):
        super(ReLUINSConvTranspose2d, self).__init__()
        self.model = nn.Sequential(ConvTranspose2d(n_in, n_out, kernel_size, stride, padding, output_padding),
                                   _Fused('InstanceNorm2d'), _Fused('ReLU'))
        self.model.apply(gaussian_weights_init)

    def forward(self, x):
        c = self.model[0]
        h = ops.conv_transpose2d(x, c.weight, None, c.stride, c.padding, c.output_padding)
        return ops.act(ops.instance_norm_(h, None, -1.0), ACT_LRELU, 0.0)


class GaussianVAE2D(nn.Module):
    """Convolutional mu / softplus(sd) heads (common_net.py:67-90)."""

    def __init__(self, n_in, n_out, kernel_size, stride, padding=0):
        super(GaussianVAE2D, self).__init__()
        self.en_mu = Conv2d(n_in, n_out, kernel_size, stride, padding)
        self.en_sigma = Conv2d(n_in, n_out, kernel_size, stride, padding)
        for m in (self.en_mu, self.en_sigma):
            m.weight.data.normal_(0, 0.002)
            m.bias.data.normal_(0, 0.002)

    def forward(self, x):
        return self.en_mu(x), ops.act(self.en_sigma(x), ops.ACT_SOFTPLUS)

    def sample(self, x, noise=None):
        mu, sd = self.forward(x)
        if noise is None:
            noise = torch.randn(mu.size(), device=mu.device, dtype=mu.dtype)
        return mu + sd * noise, mu, sd


def gaussian_kernel_1d(ksize):
    """cv2.getGaussianKernel(ksize, -1) (OpenCV's published rule: fixed tables for ksize <= 7, otherwise
    sigma = 0.3*((ksize-1)*0.5 - 1) + 0.8, normalised exp(-(i - (ksize-1)/2)^2 / (2 sigma^2)))."""
    small = {1: [1.0], 3: [0.25, 0.5, 0.25], 5: [0.0625, 0.25, 0.375, 0.25, 0.0625],
             7: [0.03125, 0.109375, 0.21875, 0.28125, 0.21875, 0.109375, 0.03125]}
    if ksize in small:
        return np.asarray(small[ksize], np.float64)
    sigma = 0.3 * ((ksize - 1) * 0.5 - 1) + 0.8
    i = np.arange(ksize, dtype=np.float64) - (ksize - 1) * 0.5
    k = np.exp(-(i * i) / (2.0 * sigma * sigma))
    return k / k.sum()


class GaussianSmoother(nn.Module):
    """Replicate-padded Gaussian blur of a 1-channel image (common_net.py:12-30); the kernel is a constant, not a
    parameter.  The blur itself is the single-input-channel conv kernel of the stems."""

    def __init__(self, kernel_size=5):
        super(GaussianSmoother, self).__init__()
        self.sigma = 0.3 * ((kernel_size - 1) * 0.5 - 1) + 0.8
        k = gaussian_kernel_1d(kernel_size)
        self.pad = (kernel_size - 1) // 2
        self.blur_kernel = torch.from_numpy(np.outer(k, k)).float().reshape(1, 1, kernel_size, kernel_size)
        self._k1d = torch.from_numpy(k).float()

    def forward(self, x):
        out = nn.functional.pad(x, [self.pad, self.pad, self.pad, self.pad], mode='replicate')
        ks = self.blur_kernel.shape[-1]
        if ks * ks <= 49:                                 # the conv kernels take up to 49 taps
            return ops.conv2d(out, self.blur_kernel.to(x.device), None, 1, 0)
        k = self._k1d.to(x.device)                        # larger kernels: the blur is separable (outer product)
        return ops.conv2d(ops.conv2d(out, k.reshape(1, 1, ks, 1).contiguous(), None, 1, 0),
                          k.reshape(1, 1, 1, ks).contiguous(), None, 1, 0)

    def cuda(self, gpu=None):
        self.blur_kernel = self.blur_kernel.cuda(gpu)
        return self
