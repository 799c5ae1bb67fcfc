"""torch.autograd.Function wrappers over the C-ABI kernels (include/lsps_hip.h).

PyTorch is used here for device memory, streams and the autograd tape only; every arithmetic
op below runs in liblsps_hip.so.  Each Function names the torch built-in of the reference it
replaces.  Inputs must be HIP float32 tensors — there is no CPU fallback (``_lib.ptr`` raises).
"""
import ctypes as _ctypes

import torch

from . import _lib
from . import options  # noqa: F401  (`ops.options.get()`: the process options in force)
from ._lib import ACT_LRELU, ACT_NONE, ACT_SOFTPLUS, ACT_TANH, LOSS_KLSD, LOSS_L1, LOSS_L2, LOSS_SQ  # noqa: F401

LRELU_SLOPE = 0.01     # nn.LeakyReLU() default (src/trainers/common_net.py:169,252,264)
IN_EPS = 1e-5          # nn.InstanceNorm2d default


def _c(t):
    return t if t.is_contiguous() else t.contiguous()


class _Span(object):
    """Brackets ONE C-ABI conv call.  The key is the kernel name the library reports for that call (lsps_last_kernel):
    nothing here mirrors the dispatch rules.  `name`: entries with a single kernel that does not go through that
    dispatcher (csrc/chwn.hip)."""

    def __init__(self, prof, flops, name=None):
        self.prof, self.flops, self.name = prof, flops, name

    def __enter__(self):
        p = self.prof
        self.timed = False
        if p.enabled or p.log is not None:
            _lib.lib().lsps_last_kernel(None)       # reset the launch counter of this thread
        if p.enabled:
            # `only`: events just around the calls that dispatched to that kernel in the learned step (same position in
            # the step's call sequence); every span still reports its kernel name, so a drifting sequence is noticed
            self.idx = p.idx
            p.idx += 1
            self.timed = p.only is None or (self.idx < len(p.learned) and p.learned[self.idx] == p.only)
            if self.timed:
                self.e0 = torch.cuda.Event(enable_timing=True)
                self.e1 = torch.cuda.Event(enable_timing=True)
                self.e0.record()        # torch's current stream == the stream the kernels are launched on

    def __exit__(self, *a):
        p = self.prof
        if not (p.enabled or p.log is not None) or a[0] is not None:
            return
        n = _ctypes.c_int(0)
        name = _lib.lib().lsps_last_kernel(_ctypes.byref(n)).decode()
        if self.name is not None:
            name, n = self.name, _ctypes.c_int(1)
        if p.log is not None:
            p.log.append(name)
        if p.enabled:
            if self.timed:
                self.e1.record()
                p.records.append((name, self.e0, self.e1, self.flops, max(n.value, 1)))
            p.step_names.append(name)
            if p.only is not None and (self.idx >= len(p.learned) or p.learned[self.idx] != name):
                p.drift += 1


class Profiler(object):
    """HIP-event timing of the conv kernels inside a timed region (bench.py `roofline`).  A span brackets
    one C-ABI conv call on the launch stream; `flops` are ALGORITHMIC (2*N*K*P*Q*C*R*S)."""

    def __init__(self):
        self.enabled = False
        self.records = []
        self.log = None                 # kernel_log_begin(): list of dispatched kernel names
        self.only = None                # restrict the events to the calls of ONE kernel (see step_begin / learn)
        self.learned = []               # kernel name per span position of one step
        self.step_names = []
        self.idx = 0
        self.drift = 0                  # spans whose kernel differed from the learned sequence while `only` was set

    def reset(self):
        self.records = []
        self.drift = 0

    def step_begin(self):
        """Marks the start of one step (a fixed sequence of conv calls).  The names seen during the previous step become
        the learned sequence unless a restriction is active."""
        if self.only is None and self.step_names:
            self.learned = self.step_names
        self.step_names = []
        self.idx = 0

    def restrict_to(self, kernel):
        """Timed regions pay ~2.5 us of launch-stream time per recorded event (1.3 % of the bs=128 step with ~600 spans):
        keep the events of ONE kernel's calls only (the dominant one, which `roofline` is about); None lifts it."""
        if kernel is not None and self.step_names:
            self.learned = self.step_names
            self.step_names = []
        self.only = kernel

    def span(self, flops, name=None):
        return _Span(self, flops, name)

    def summary(self):
        torch.cuda.synchronize()
        agg = {}
        for key, e0, e1, flops, launches in self.records:
            a = agg.setdefault(key, dict(calls=0, launches=0, total_ms=0.0, flop=0.0))
            a['calls'] += 1
            a['launches'] += launches
            a['total_ms'] += e0.elapsed_time(e1)
            a['flop'] += flops
        out = {}
        for key, a in agg.items():
            out[key] = dict(calls=a['calls'], launches=a['launches'], total_ms=a['total_ms'],
                            avg_ms=a['total_ms'] / max(a['launches'], 1),
                            gflop_per_launch=a['flop'] / max(a['launches'], 1) / 1e9,
                            tflops=a['flop'] / max(a['total_ms'], 1e-9) / 1e9)
        return out


profiler = Profiler()


def kernel_log_begin():
    """Starts recording the name of the kernel every conv call dispatches to (as reported by the library)."""
    profiler.log = []


def kernel_log_end():
    names, profiler.log = profiler.log or [], None
    return names


def set_math_mode(mode):
    """'f32' (default, exact f32 MFMA) or 'bf16' (bf16 MFMA operands, f32 accumulation, in the 3x3 residual-conv
    kernels: BASELINE config 5).  Process-wide."""
    code = {'f32': 0, 'bf16': 1, 'f32_split': 2}[mode]
    _lib.check(_lib.lib().lsps_set_math_mode(code), 'set_math_mode')


def get_math_mode():
    return ('f32', 'bf16', 'f32_split')[_lib.lib().lsps_get_math_mode()]


_WINO_MODES = ('off', 'auto', 'always', 'auto_f2', 'always_f2')


def set_winograd(mode):
    """Algorithm of the f32 3x3 / stride-1 / width-32 convs: 'off' (direct implicit GEMM), 'auto' (Winograd when the grid
    fills the chip: F(4x4,3x3) on 32x32 maps, else F(2x2,3x3); default), 'always' (every eligible shape), 'auto_f2' /
    'always_f2' (the same without the F(4x4,3x3) kernel).  Process-wide."""
    _lib.check(_lib.lib().lsps_set_winograd(_WINO_MODES.index(mode)), 'set_winograd')


def get_winograd():
    return _WINO_MODES[_lib.lib().lsps_get_winograd()]


def conv_out_size(h, r, stride, pad):
    return (h + 2 * pad - r) // stride + 1


def convT_out_size(h, r, stride, pad, outpad):
    return (h - 1) * stride - 2 * pad + r + outpad


# ------------------------------------------------------------------------------------------
# Conv2d (+ fused bias and LeakyReLU/Tanh epilogue)
# ------------------------------------------------------------------------------------------
class _EmptyBatchFn(torch.autograd.Function):
    """Empty batch in -> empty batch out, as torch.nn does (the kernels are never launched on zero samples); the
    parameter gradients of an empty batch are zeros."""

    @staticmethod
    def forward(ctx, x, out_shape, *params):
        ctx.xshape = tuple(x.shape)
        ctx.pshapes = [None if p is None else tuple(p.shape) for p in params]
        ctx.dev = x.device
        return torch.empty(out_shape, dtype=torch.float32, device=x.device)

    @staticmethod
    def backward(ctx, g):
        gx = torch.empty(ctx.xshape, dtype=torch.float32, device=ctx.dev)
        gp = [None if s is None else torch.zeros(s, dtype=torch.float32, device=ctx.dev) for s in ctx.pshapes]
        return (gx, None) + tuple(gp)


def _empty(x, out_shape, *params):
    return _EmptyBatchFn.apply(x, tuple(out_shape), *params)


class ActHolder(object):
    """Shared by a C8 layer with a fused LeakyReLU epilogue (the producer) and the ONE layer that consumes its output
    (common_net.run_layers pairs consecutive layers): the consumer's dgrad kernel multiplies the gradient it hands back by
    LeakyReLU'(producer output) in its epilogue and leaves the producer's bias gradient here, so the producer's own
    activation-backward pass (lsps_c8_act_bwd_bias / lsps_act_bwd_bias: three passes over the layer's output) is skipped."""
    __slots__ = ('slope', 'fused', 'db', 'grad', 'want')

    def __init__(self, slope, want=None):
        # `want` / `grad` (round 5): the producer and the consumer keep their activations in DIFFERENT layouts (f32 NCHW vs
        # three-limb X3), so the gradient the consumer's fused epilogue produces cannot travel along the autograd edge (its shape
        # is the consumer's input's).  The producer announces the layout it needs ('x3' | 'f32'); the consumer's backward writes
        # it, leaves it in `grad` and returns None for that input; the producer's backward (set_materialize_grads(False): it
        # receives None) picks it up here.
        self.slope, self.fused, self.db, self.grad, self.want = float(slope), False, None, None, want


def _fuse_enabled():
    return options.get().fuse_act


def x3_fuse_enabled():
    """ONE predicate for both sides of a fused LeakyReLU backward on C8 / X3 tensors (the X3 stem paired with its
    consumer on `fuse_act` alone while the consumer's backward also asked for `c8_fuse_act`)."""
    o = options.get()
    return o.c8_fuse_act and o.fuse_act


def _fusable(prev):
    return prev is not None and prev.slope >= 0 and x3_fuse_enabled()


def _act_backward(L, dy, y, act, slope, want_db, channels, ws, wsb, st):
    """Gradient through the fused output activation of a conv / transposed conv.  When the layer's bias gradient is
    wanted too it comes out of the same pass (db = sum over n, h, w of the pre-activation gradient)."""
    if act == ACT_NONE:
        return dy, None
    dpre = torch.empty_like(dy)
    if want_db:
        db = torch.empty(channels, dtype=torch.float32, device=dy.device)
        _lib.check(L.lsps_act_bwd_bias(_lib.ptr(dy), _lib.ptr(y), _lib.ptr(dpre), _lib.ptr(db), dy.shape[0], channels,
                                       dy.shape[2] * dy.shape[3], act, slope, ws, wsb, st), 'act_bwd_bias')
        return dpre, db
    _lib.check(L.lsps_act_bwd(_lib.ptr(dy), _lib.ptr(y), _lib.ptr(dpre), dy.numel(), act, slope, st), 'act_bwd')
    return dpre, None


class _Conv2dFn(torch.autograd.Function):
    """nn.Conv2d [+ nn.LeakyReLU(inplace)] — common_net.py:250-252, 162-163; lsps_nets.py:123-124."""

    @staticmethod
    def forward(ctx, x, w, b, stride, pad, act, slope):
        L = _lib.lib()
        x, w = _c(x), _c(w)
        N, C, H, W = x.shape
        K, C2, R, S = w.shape
        assert C == C2, "channel mismatch"
        P, Q = conv_out_size(H, R, stride, pad), conv_out_size(W, S, stride, pad)
        y = torch.empty((N, K, P, Q), dtype=torch.float32, device=x.device)
        ws, wsb = _lib.workspace(L.lsps_conv2d_workspace_bytes(N, C, H, W, K, R, S, stride, pad), x.device)
        with profiler.span(2.0 * N * K * P * Q * C * R * S):
            _lib.check(L.lsps_conv2d_fwd(_lib.ptr(x), _lib.ptr(w), _lib.ptr(b), _lib.ptr(y), N, C, H, W, K, R, S,
                                         stride, pad, act, slope, ws, wsb, _lib.stream()), 'conv2d_fwd')
        ctx.geom = (N, C, H, W, K, R, S, stride, pad, act, slope)
        ctx.has_bias = b is not None
        ctx.save_for_backward(x, w, y if act != ACT_NONE else None)
        return y

    @staticmethod
    def backward(ctx, dy):
        L = _lib.lib()
        x, w, y = ctx.saved_tensors
        N, C, H, W, K, R, S, stride, pad, act, slope = ctx.geom
        dy = _c(dy)
        st = _lib.stream()
        flops = 2.0 * N * K * dy.shape[2] * dy.shape[3] * C * R * S
        ws, wsb = _lib.workspace(L.lsps_conv2d_workspace_bytes(N, C, H, W, K, R, S, stride, pad), x.device)
        dx = dw = db = None
        want_db = ctx.has_bias and ctx.needs_input_grad[2]
        if C == 1 and act == ACT_LRELU and slope >= 0 and not ctx.needs_input_grad[0] and (ctx.needs_input_grad[1] or want_db) and \
                _fuse_enabled() and L.lsps_conv2d_stem_wgrad_act_ok(N, H, W, K, R, S, stride, pad) == 1:
            # one-input-channel stem: the LeakyReLU backward is applied while dy is staged and the bias gradient is one more
            # column of the same product — no pass over the net's largest activation (csrc/conv_c1.h)
            dw = torch.empty_like(w)
            db = torch.empty(K, dtype=torch.float32, device=x.device) if want_db else None
            with profiler.span(flops):
                _lib.check(L.lsps_conv2d_stem_wgrad_act(_lib.ptr(x), _lib.ptr(dy), _lib.ptr(y), _lib.ptr(dw), _lib.ptr(db), N, H, W, K, R,
                                                        S, stride, pad, slope, ws, wsb, st), 'conv2d_stem_wgrad_act')
            return None, dw, db, None, None, None, None
        dy, db = _act_backward(L, dy, y, act, slope, want_db, K, ws, wsb, st)
        if ctx.needs_input_grad[0]:
            dx = torch.empty_like(x)
            with profiler.span(flops):
                _lib.check(L.lsps_conv2d_dgrad(_lib.ptr(dy), _lib.ptr(w), _lib.ptr(dx), N, C, H, W, K, R, S, stride,
                                               pad, ws, wsb, st), 'conv2d_dgrad')
        if ctx.needs_input_grad[1] or (ctx.has_bias and ctx.needs_input_grad[2]):
            dw = torch.empty_like(w)
            db_here = None                  # the bias gradient, unless the fused act_bwd pass already produced it
            if ctx.has_bias and ctx.needs_input_grad[2] and db is None:
                db = db_here = torch.empty(K, dtype=torch.float32, device=x.device)
            with profiler.span(flops):
                _lib.check(L.lsps_conv2d_wgrad(_lib.ptr(x), _lib.ptr(dy), _lib.ptr(dw), _lib.ptr(db_here), N, C, H, W, K,
                                               R, S, stride, pad, ws, wsb, st), 'conv2d_wgrad')
        return dx, dw, db, None, None, None, None


def conv2d(x, w, b=None, stride=1, pad=0, act=ACT_NONE, slope=LRELU_SLOPE):
    if x.shape[0] == 0:
        return _empty(x, (0, w.shape[0], conv_out_size(x.shape[2], w.shape[2], stride, pad),
                          conv_out_size(x.shape[3], w.shape[3], stride, pad)), w, b)
    return _Conv2dFn.apply(x, w, b, int(stride), int(pad), int(act), float(slope))


class _Conv2dGroupedFn(torch.autograd.Function):
    """nn.Conv2d(..., groups=G) — the 3x3 conv of LeakyINSResNeXtBlock (common_net.py:116).  One launch per group on the
    channel slices of the full tensors (lsps_conv2d_grouped_*): no slice copies, no concatenation."""

    @staticmethod
    def forward(ctx, x, w, b, stride, pad, groups):
        L = _lib.lib()
        x, w = _c(x), _c(w)
        N, C, H, W = x.shape
        K, Cg, R, S = w.shape
        assert C == Cg * groups and K % groups == 0, "channel / group mismatch"
        P, Q = conv_out_size(H, R, stride, pad), conv_out_size(W, S, stride, pad)
        y = torch.empty((N, K, P, Q), dtype=torch.float32, device=x.device)
        ws, wsb = _lib.workspace(L.lsps_conv2d_workspace_bytes(N, Cg, H, W, K // groups, R, S, stride, pad), x.device)
        with profiler.span(2.0 * N * K * P * Q * Cg * R * S):
            _lib.check(L.lsps_conv2d_grouped_fwd(_lib.ptr(x), _lib.ptr(w), _lib.ptr(b), _lib.ptr(y), N, C, H, W, K, R, S, stride,
                                                 pad, groups, ACT_NONE, 1.0, ws, wsb, _lib.stream()), 'conv2d_grouped_fwd')
        ctx.geom = (N, C, H, W, K, R, S, stride, pad, groups)
        ctx.has_bias = b is not None
        ctx.save_for_backward(x, w)
        return y

    @staticmethod
    def backward(ctx, dy):
        L = _lib.lib()
        x, w = ctx.saved_tensors
        N, C, H, W, K, R, S, stride, pad, groups = ctx.geom
        dy = _c(dy)
        st = _lib.stream()
        flops = 2.0 * N * K * dy.shape[2] * dy.shape[3] * (C // groups) * R * S
        ws, wsb = _lib.workspace(L.lsps_conv2d_workspace_bytes(N, C // groups, H, W, K // groups, R, S, stride, pad), x.device)
        dx = dw = db = None
        if ctx.needs_input_grad[0]:
            dx = torch.empty_like(x)
            with profiler.span(flops):
                _lib.check(L.lsps_conv2d_grouped_dgrad(_lib.ptr(dy), _lib.ptr(w), _lib.ptr(dx), N, C, H, W, K, R, S, stride, pad,
                                                       groups, ws, wsb, st), 'conv2d_grouped_dgrad')
        if ctx.needs_input_grad[1] or (ctx.has_bias and ctx.needs_input_grad[2]):
            dw = torch.empty_like(w)
            if ctx.has_bias and ctx.needs_input_grad[2]:
                db = torch.empty(K, dtype=torch.float32, device=x.device)
            with profiler.span(flops):
                _lib.check(L.lsps_conv2d_grouped_wgrad(_lib.ptr(x), _lib.ptr(dy), _lib.ptr(dw), _lib.ptr(db), N, C, H, W, K, R, S,
                                                       stride, pad, groups, ws, wsb, st), 'conv2d_grouped_wgrad')
        return dx, dw, db, None, None, None


def conv2d_grouped(x, w, b=None, stride=1, pad=0, groups=1):
    if groups == 1:
        return conv2d(x, w, b, stride, pad)
    if x.shape[0] == 0:
        return _empty(x, (0, w.shape[0], conv_out_size(x.shape[2], w.shape[2], stride, pad),
                          conv_out_size(x.shape[3], w.shape[3], stride, pad)), w, b)
    return _Conv2dGroupedFn.apply(x, w, b, int(stride), int(pad), int(groups))


# ------------------------------------------------------------------------------------------
# ConvTranspose2d
# ------------------------------------------------------------------------------------------
class _ConvT2dFn(torch.autograd.Function):
    """nn.ConvTranspose2d [+ LeakyReLU / Tanh] — common_net.py:262-264; lsps_nets.py:17-23, 226-229."""

    @staticmethod
    def forward(ctx, x, w, b, stride, pad, outpad, act, slope, prev=None, own=None):
        L = _lib.lib()
        x, w = _c(x), _c(w)
        N, Ci, H, W = x.shape
        Ci2, Co, R, S = w.shape
        ctx.prev, ctx.own = prev, own
        assert Ci == Ci2, "channel mismatch"
        Ho, Wo = convT_out_size(H, R, stride, pad, outpad), convT_out_size(W, S, stride, pad, outpad)
        y = torch.empty((N, Co, Ho, Wo), dtype=torch.float32, device=x.device)
        ws, wsb = _lib.workspace(L.lsps_convT2d_workspace_bytes(N, Ci, H, W, Co, R, S, stride, pad, outpad), x.device)
        with profiler.span(2.0 * N * Ci * H * W * Co * R * S):
            _lib.check(L.lsps_convT2d_fwd(_lib.ptr(x), _lib.ptr(w), _lib.ptr(b), _lib.ptr(y), N, Ci, H, W, Co, R, S,
                                          stride, pad, outpad, act, slope, ws, wsb, _lib.stream()), 'convT2d_fwd')
        ctx.geom = (N, Ci, H, W, Co, R, S, stride, pad, outpad, act, slope)
        ctx.has_bias = b is not None
        ctx.save_for_backward(x, w, y if act != ACT_NONE else None)
        return y

    @staticmethod
    def backward(ctx, dy):
        L = _lib.lib()
        x, w, y = ctx.saved_tensors
        N, Ci, H, W, Co, R, S, stride, pad, outpad, act, slope = ctx.geom
        dy = _c(dy)
        st = _lib.stream()
        flops = 2.0 * N * Ci * H * W * Co * R * S
        ws, wsb = _lib.workspace(L.lsps_convT2d_workspace_bytes(N, Ci, H, W, Co, R, S, stride, pad, outpad), x.device)
        dx = dw = db = None
        want_db = ctx.has_bias and ctx.needs_input_grad[2]
        if ctx.own is not None and ctx.own.fused:        # the consumer's dgrad already applied this layer's LeakyReLU'
            db = ctx.own.db if want_db else None
            ctx.own.fused, ctx.own.db = False, None
        else:
            dy, db = _act_backward(L, dy, y, act, slope, want_db, Co, ws, wsb, st)
        if ctx.needs_input_grad[0]:
            dx = torch.empty_like(x)
            with profiler.span(flops):
                if _fusable(ctx.prev) and (R, S, Co, stride, pad, outpad) == (1, 1, 1, 1, 0, 0) and Ci <= 64 and (H * W) % 4 == 0:
                    # 1x1 output head: its input is the previous layer's output; that layer's LeakyReLU backward rides along
                    dbp = torch.empty(Ci, dtype=torch.float32, device=x.device)
                    need_w = ctx.needs_input_grad[1]                     # the head's own weight gradient comes out of the same pass
                    if need_w:                                           # (its bias gradient: from the activation backward above)
                        dw = torch.empty_like(w)
                    wsd, wsdb = _lib.workspace(L.lsps_pw1_dgrad_act_workspace_bytes(N, Ci), x.device)
                    if ctx.prev.want == 'x3' and Ci % 16 == 0:           # the layer in front is a three-limb layer: hand its gradient
                        dx = None
                        gl = torch.empty((N, 3, Ci // 8, H, W, 8), dtype=BF16, device=x.device)       # over as X3 (ActHolder.grad)
                        _lib.check(L.lsps_pw1_dgrad_act_x3(_lib.ptr(dy), _lib.ptr(w), _lib.ptr(x), ctx.prev.slope, _lib.ptr(gl, BF16),
                                                           _lib.ptr(dbp), _lib.ptr(dw), None, N, Ci, H * W, wsd, wsdb, st), 'pw1_dgrad_act_x3')
                        ctx.prev.grad, dx = gl, None
                    else:
                        _lib.check(L.lsps_pw1_dgrad_act(_lib.ptr(dy), _lib.ptr(w), _lib.ptr(x), ctx.prev.slope, _lib.ptr(dx), _lib.ptr(dbp),
                                                        _lib.ptr(dw), None, N, Ci, H * W, wsd, wsdb, st), 'pw1_dgrad_act')
                    ctx.prev.fused, ctx.prev.db = True, dbp
                    if need_w and (db is not None or not want_db):
                        return dx, dw, db, None, None, None, None, None, None, None
                else:
                    _lib.check(L.lsps_convT2d_dgrad(_lib.ptr(dy), _lib.ptr(w), _lib.ptr(dx), N, Ci, H, W, Co, R, S, stride,
                                                    pad, outpad, ws, wsb, st), 'convT2d_dgrad')
        if ctx.needs_input_grad[1] or (ctx.has_bias and ctx.needs_input_grad[2]):
            dw = torch.empty_like(w)
            db_here = None
            if ctx.has_bias and ctx.needs_input_grad[2] and db is None:
                db = db_here = torch.empty(Co, dtype=torch.float32, device=x.device)
            with profiler.span(flops):
                _lib.check(L.lsps_convT2d_wgrad(_lib.ptr(x), _lib.ptr(dy), _lib.ptr(dw), _lib.ptr(db_here), N, Ci, H, W, Co,
                                                R, S, stride, pad, outpad, ws, wsb, st), 'convT2d_wgrad')
        return dx, dw, db, None, None, None, None, None, None, None


def conv_transpose2d(x, w, b=None, stride=1, pad=0, outpad=0, act=ACT_NONE, slope=LRELU_SLOPE, prev=None, own=None):
    """`own` / `prev`: ActHolder of this layer / of the layer whose output `x` is (see ActHolder)."""
    if x.shape[0] == 0:
        return _empty(x, (0, w.shape[1], convT_out_size(x.shape[2], w.shape[2], stride, pad, outpad),
                          convT_out_size(x.shape[3], w.shape[3], stride, pad, outpad)), w, b)
    return _ConvT2dFn.apply(x, w, b, int(stride), int(pad), int(outpad), int(act), float(slope), prev, own)


# ------------------------------------------------------------------------------------------
# 3x3 / stride-2 convs on small maps in batch-innermost layout [C][H][W][N] (the discriminator trunk; csrc/chwn.hip)
# ------------------------------------------------------------------------------------------
class _Transpose2dFn(torch.autograd.Function):
    """[R][S] -> [S][R] of a contiguous tensor viewed as a matrix (NCHW <-> CHWN); backward = the inverse transpose."""

    @staticmethod
    def forward(ctx, x, R, S, out_shape):
        x = _c(x)
        y = torch.empty(out_shape, dtype=torch.float32, device=x.device)
        _lib.check(_lib.lib().lsps_transpose2d(_lib.ptr(x), _lib.ptr(y), R, S, _lib.stream()), 'transpose2d')
        ctx.dims = (R, S, tuple(x.shape))
        return y

    @staticmethod
    def backward(ctx, g):
        R, S, in_shape = ctx.dims
        g = _c(g)
        dx = torch.empty(in_shape, dtype=torch.float32, device=g.device)
        _lib.check(_lib.lib().lsps_transpose2d(_lib.ptr(g), _lib.ptr(dx), S, R, _lib.stream()), 'transpose2d')
        return dx, None, None, None


def nchw_to_chwn(x):
    N, C, H, W = x.shape
    return _Transpose2dFn.apply(x, N, C * H * W, (C, H, W, N))


def chwn_to_nchw(x):
    C, H, W, N = x.shape
    return _Transpose2dFn.apply(x, C * H * W, N, (N, C, H, W))


def conv3x3s2_chwn_ok(N, C, H, W, K):
    return _lib.lib().lsps_conv3x3s2_chwn_workspace_bytes(N, C, H, W, K) > 0 and get_math_mode() == 'f32'


class _ConvS2CHWNFn(torch.autograd.Function):
    """LeakyReLUConv2d(C, K, 3, 2, 1) (common_net.py:250-252) on x [C][H][W][N] -> [K][H/2][W/2][N]."""

    @staticmethod
    def forward(ctx, x, w, b, act, slope):
        L = _lib.lib()
        x, w = _c(x), _c(w)
        C, H, W, N = x.shape
        K = w.shape[0]
        assert tuple(w.shape) == (K, C, 3, 3), "conv3x3s2_chwn: weight must be (K, C, 3, 3)"
        y = torch.empty((K, H // 2, W // 2, N), dtype=torch.float32, device=x.device)
        ws, wsb = _lib.workspace(L.lsps_conv3x3s2_chwn_workspace_bytes(N, C, H, W, K), x.device)
        with profiler.span(2.0 * N * K * (H // 2) * (W // 2) * C * 9, 'chwn_gemm_kernel'):
            _lib.check(L.lsps_conv3x3s2_chwn_fwd(_lib.ptr(x), _lib.ptr(w), _lib.ptr(b), _lib.ptr(y), N, C, H, W, K, act, slope,
                                                 ws, wsb, _lib.stream()), 'conv3x3s2_chwn_fwd')
        ctx.geom = (N, C, H, W, K, act, slope)
        ctx.has_bias = b is not None
        ctx.save_for_backward(x, w, y if act != ACT_NONE else None)
        return y

    @staticmethod
    def backward(ctx, dy):
        L = _lib.lib()
        x, w, y = ctx.saved_tensors
        N, C, H, W, K, act, slope = ctx.geom
        dy = _c(dy)
        st = _lib.stream()
        P, Q = H // 2, W // 2
        flops = 2.0 * N * K * P * Q * C * 9
        ws, wsb = _lib.workspace(max(L.lsps_conv3x3s2_chwn_workspace_bytes(N, C, H, W, K), 1 << 20), x.device)
        dx = dw = db = None
        # the layout makes the layer look like one image with P*Q*N pixels per channel to the activation-backward kernels
        if act != ACT_NONE:
            dpre = torch.empty_like(dy)
            if ctx.has_bias and ctx.needs_input_grad[2]:
                db = torch.empty(K, dtype=torch.float32, device=dy.device)
                _lib.check(L.lsps_act_bwd_bias(_lib.ptr(dy), _lib.ptr(y), _lib.ptr(dpre), _lib.ptr(db), 1, K, P * Q * N, act, slope,
                                               ws, wsb, st), 'act_bwd_bias')
            else:
                _lib.check(L.lsps_act_bwd(_lib.ptr(dy), _lib.ptr(y), _lib.ptr(dpre), dy.numel(), act, slope, st), 'act_bwd')
            dy = dpre
        elif ctx.has_bias and ctx.needs_input_grad[2]:
            db = dy.reshape(K, -1).sum(1)
        if ctx.needs_input_grad[0]:
            dx = torch.empty_like(x)
            with profiler.span(flops, 'chwn_gemm_kernel'):
                _lib.check(L.lsps_conv3x3s2_chwn_dgrad(_lib.ptr(dy), _lib.ptr(w), _lib.ptr(dx), N, C, H, W, K, ws, wsb, st),
                           'conv3x3s2_chwn_dgrad')
        if ctx.needs_input_grad[1]:
            dw = torch.empty_like(w)
            with profiler.span(flops, 'chwn_wgrad_kernel'):
                _lib.check(L.lsps_conv3x3s2_chwn_wgrad(_lib.ptr(x), _lib.ptr(dy), _lib.ptr(dw), N, C, H, W, K, ws, wsb, st),
                           'conv3x3s2_chwn_wgrad')
        return dx, dw, db, None, None


def conv3x3s2_chwn(x, w, b=None, act=ACT_NONE, slope=LRELU_SLOPE):
    return _ConvS2CHWNFn.apply(x, w, b, int(act), float(slope))


# ------------------------------------------------------------------------------------------
# InstanceNorm2d(affine=False) [+ LeakyReLU] [+ residual], in place on the conv output
# ------------------------------------------------------------------------------------------
class _InormFn(torch.autograd.Function):
    """nn.InstanceNorm2d (+ nn.LeakyReLU(inplace) | `out += residual`) — common_net.py:168-171, 177-181.
    Works in place on ``y`` (a fresh conv output nobody else saved); the backward is computed from the
    saved OUTPUT, so the pre-norm tensor is never kept."""

    @staticmethod
    def forward(ctx, y, residual, slope):
        L = _lib.lib()
        assert y.is_contiguous()
        N, C, H, W = y.shape
        if residual is not None:
            residual = _c(residual)
        rstd = torch.empty(N * C, dtype=torch.float32, device=y.device)
        _lib.check(L.lsps_inorm_fwd(_lib.ptr(y), _lib.ptr(residual), _lib.ptr(y), _lib.ptr(rstd), N * C, H * W,
                                    IN_EPS, slope, _lib.stream()), 'inorm_fwd')
        ctx.mark_dirty(y)
        ctx.slope = slope
        ctx.has_res = residual is not None
        ctx.save_for_backward(y, residual, rstd)
        return y

    @staticmethod
    def backward(ctx, dout):
        L = _lib.lib()
        out, residual, rstd = ctx.saved_tensors
        dout = _c(dout)
        N, C, H, W = out.shape
        dy = torch.empty_like(out)
        _lib.check(L.lsps_inorm_bwd(_lib.ptr(dout), _lib.ptr(out), _lib.ptr(residual), _lib.ptr(rstd), _lib.ptr(dy),
                                    N * C, H * W, ctx.slope, _lib.stream()), 'inorm_bwd')
        return dy, (dout if ctx.has_res and ctx.needs_input_grad[1] else None), None


class _ResBlockFn(torch.autograd.Function):
    """LeakyINSResBlock as ONE autograd node: x + IN(conv3x3(LReLU(IN(conv3x3(x))))) (common_net.py:160-181, stride 1).
    Same kernels as the composed form; what the fusion buys is the backward: the gradient arriving over the skip
    connection is added in the epilogue of the first conv's dgrad kernel (`lsps_conv2d_dgrad_acc`) instead of a separate
    3-pass add by autograd, and five saved-tensor / node hand-offs disappear."""

    @staticmethod
    def forward(ctx, x, w1, w2, nograd=False):
        L = _lib.lib()
        x, w1, w2 = _c(x), _c(w1), _c(w2)
        N, C, H, W = x.shape
        K = w1.shape[0]
        assert w1.shape == (K, C, 3, 3) and w2.shape == (K, K, 3, 3) and K == C, "residual block: C -> C, 3x3"
        st = _lib.stream()
        ws, wsb = _lib.workspace(L.lsps_conv2d_workspace_bytes(N, C, H, W, K, 3, 3, 1, 1), x.device)
        flops = 2.0 * N * K * H * W * C * 9
        a1 = torch.empty((N, K, H, W), dtype=torch.float32, device=x.device)
        y = torch.empty_like(a1)
        r1 = torch.empty(N * K, dtype=torch.float32, device=x.device)
        r2 = torch.empty_like(r1)
        # conv + InstanceNorm (+ LeakyReLU | + skip) in ONE call each: on 32x32 maps the Winograd F(4x4,3x3) kernel owns
        # whole (n, k) planes and normalises in its epilogue; other shapes run conv + the in-place norm pass in the library
        # a pass nobody differentiates (the caller saw torch.no_grad(): inside forward() grad mode is always off, and
        # needs_input_grad follows the weights' requires_grad whatever the mode) may take the few-image dispatch of
        # lsps_conv2d_in_fwd_nograd (include/lsps_hip.h)
        in_fwd = L.lsps_conv2d_in_fwd_nograd if nograd else L.lsps_conv2d_in_fwd
        with profiler.span(flops):
            _lib.check(in_fwd(_lib.ptr(x), _lib.ptr(w1), None, _lib.ptr(a1), _lib.ptr(r1), N, C, H, W, K,
                              LRELU_SLOPE, IN_EPS, ws, wsb, st), 'conv2d_in_fwd')
        with profiler.span(flops):
            _lib.check(in_fwd(_lib.ptr(a1), _lib.ptr(w2), _lib.ptr(x), _lib.ptr(y), _lib.ptr(r2), N, K, H, W, K,
                              -1.0, IN_EPS, ws, wsb, st), 'conv2d_in_fwd')
        ctx.save_for_backward(x, w1, w2, a1, y, r1, r2)
        return y

    @staticmethod
    def backward(ctx, g):
        L = _lib.lib()
        x, w1, w2, a1, y, r1, r2 = ctx.saved_tensors
        g = _c(g)
        N, C, H, W = x.shape
        K = w1.shape[0]
        st = _lib.stream()
        ws, wsb = _lib.workspace(L.lsps_conv2d_workspace_bytes(N, C, H, W, K, 3, 3, 1, 1), x.device)
        flops = 2.0 * N * K * H * W * C * 9
        dh2 = torch.empty_like(y)
        _lib.check(L.lsps_inorm_bwd(_lib.ptr(g), _lib.ptr(y), _lib.ptr(x), _lib.ptr(r2), _lib.ptr(dh2), N * K, H * W, -1.0,
                                    st), 'inorm_bwd')
        dw1 = dw2 = dx = None
        if ctx.needs_input_grad[2]:
            dw2 = torch.empty_like(w2)
            with profiler.span(flops):
                _lib.check(L.lsps_conv2d_wgrad(_lib.ptr(a1), _lib.ptr(dh2), _lib.ptr(dw2), None, N, K, H, W, K, 3, 3, 1, 1,
                                               ws, wsb, st), 'conv2d_wgrad')
        # dgrad of the second conv pushed through InstanceNorm + LeakyReLU in ONE call: on 32x32 maps the norm backward runs
        # in the epilogue of the F(4x4,3x3) dgrad kernel (da1 never reaches HBM)
        dh1 = torch.empty_like(a1)
        with profiler.span(flops):
            _lib.check(L.lsps_conv2d_dgrad_inbwd(_lib.ptr(dh2), _lib.ptr(w2), _lib.ptr(a1), _lib.ptr(r1), _lib.ptr(dh1), N, K, H,
                                                 W, K, LRELU_SLOPE, ws, wsb, st), 'conv2d_dgrad_inbwd')
        da1 = dh2                                        # dh2 is dead from here on: its storage receives dx below
        if ctx.needs_input_grad[1]:
            dw1 = torch.empty_like(w1)
            with profiler.span(flops):
                _lib.check(L.lsps_conv2d_wgrad(_lib.ptr(x), _lib.ptr(dh1), _lib.ptr(dw1), None, N, C, H, W, K, 3, 3, 1, 1, ws,
                                               wsb, st), 'conv2d_wgrad')
        if ctx.needs_input_grad[0]:
            dx = da1                                     # da1 is dead: reuse
            with profiler.span(flops):
                _lib.check(L.lsps_conv2d_dgrad_acc(_lib.ptr(dh1), _lib.ptr(w1), _lib.ptr(g), _lib.ptr(dx), N, C, H, W, K, 3, 3,
                                                   1, 1, ws, wsb, st), 'conv2d_dgrad_acc')
        return dx, dw1, dw2, None


# ------------------------------------------------------------------------------------------
# bf16 residual trunk in the channel-group layout "C8": [N][C/8][H][W][8] bf16 tensors (csrc/c8conv.h, c8wgrad.h)
# ------------------------------------------------------------------------------------------
BF16 = torch.bfloat16


def is_c8(t):
    return torch.is_tensor(t) and t.dtype == BF16 and t.dim() == 5 and t.shape[-1] == 8


def c8_block_ok(x, channels, dropout=0.0):
    """Can a LeakyINSResBlock(channels -> channels) on `x` (f32 NCHW or already C8) run on the C8 kernels?  bf16 math mode,
    32x32 maps, channels % 128 == 0 (k tiles of the weight-gradient kernel), no dropout; LSPS_C8=0 switches the path off."""
    if get_math_mode() != 'bf16' or dropout > 0 or not options.get().c8:
        return False
    if is_c8(x):
        N, G, H, W, _ = x.shape
        C = G * 8
    else:
        if x.dim() != 4:
            return False
        N, C, H, W = x.shape
    return N > 0 and C == channels and channels % 128 == 0 and H == 32 and W == 32


class _ToC8Fn(torch.autograd.Function):
    """f32 [N][C][H][W] -> bf16 [N][C/8][H][W][8]; backward = the inverse conversion of the gradient."""

    @staticmethod
    def forward(ctx, x):
        x = _c(x)
        N, C, H, W = x.shape
        y = torch.empty((N, C // 8, H, W, 8), dtype=BF16, device=x.device)
        _lib.check(_lib.lib().lsps_c8_from_nchw(_lib.ptr(x), _lib.ptr(y, BF16), N, C, H * W, _lib.stream()), 'c8_from_nchw')
        return y

    @staticmethod
    def backward(ctx, g):
        g = _c(g)
        N, G, H, W, _ = g.shape
        dx = torch.empty((N, G * 8, H, W), dtype=torch.float32, device=g.device)
        _lib.check(_lib.lib().lsps_c8_to_nchw(_lib.ptr(g, BF16), _lib.ptr(dx), N, G * 8, H * W, _lib.stream()), 'c8_to_nchw')
        return dx


class _FromC8Fn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, y):
        y = _c(y)
        N, G, H, W, _ = y.shape
        x = torch.empty((N, G * 8, H, W), dtype=torch.float32, device=y.device)
        _lib.check(_lib.lib().lsps_c8_to_nchw(_lib.ptr(y, BF16), _lib.ptr(x), N, G * 8, H * W, _lib.stream()), 'c8_to_nchw')
        return x

    @staticmethod
    def backward(ctx, g):
        g = _c(g)
        N, C, H, W = g.shape
        dy = torch.empty((N, C // 8, H, W, 8), dtype=BF16, device=g.device)
        _lib.check(_lib.lib().lsps_c8_from_nchw(_lib.ptr(g), _lib.ptr(dy, BF16), N, C, H * W, _lib.stream()), 'c8_from_nchw')
        return dy


def to_c8(x):
    return x if is_c8(x) else _ToC8Fn.apply(x)


def from_c8(x):
    """f32 NCHW from whatever layout a run_layers chain left `x` in (C8 bf16, three-limb X3, or already f32 NCHW)."""
    if is_c8(x):
        return _FromC8Fn.apply(x)
    if is_x3(x):
        return _FromX3Fn.apply(x)
    return x


class _AddC8Fn(torch.autograd.Function):
    """a + b on C8 tensors (GaussianNoiseLayer, common_net.py:39-40); b is a constant."""

    @staticmethod
    def forward(ctx, a, b):
        a, b = _c(a), _c(b)
        out = torch.empty_like(a)
        if not is_c8(b):                                 # f32 [N, C, H, W]: the draw as it was made
            N, G, H, W, _ = a.shape
            assert tuple(b.shape) == (N, G * 8, H, W)
            _lib.check(_lib.lib().lsps_c8_add_nchw(_lib.ptr(a, BF16), _lib.ptr(b), _lib.ptr(out, BF16), N, G * 8, H * W, _lib.stream()),
                       'c8_add_nchw')
            return out
        assert a.shape == b.shape
        _lib.check(_lib.lib().lsps_c8_add(_lib.ptr(a, BF16), _lib.ptr(b, BF16), _lib.ptr(out, BF16), a.numel(), _lib.stream()), 'c8_add')
        return out

    @staticmethod
    def backward(ctx, g):
        return g, None


def add_c8(a, b):
    return _AddC8Fn.apply(a, b)


def _c8_enabled():
    return get_math_mode() == 'bf16' and options.get().c8


def c8_conv_s2_ok(x, w, stride, pad):
    """Can LeakyReLUConv2d(C, K, 3, stride 2, pad 1) on `x` (f32 NCHW or C8) run on the C8 stride-2 kernels (csrc/c8s2.h)?
    bf16 math mode only; LSPS_C8=0 / LSPS_C8S2=0 switch it off."""
    if not _c8_enabled() or not options.get().c8s2 or stride != 2 or pad != 1 or tuple(w.shape[2:]) != (3, 3):
        return False
    if is_c8(x):
        N, G, H, W, _ = x.shape
        C = G * 8
    elif x.dim() == 4:
        N, C, H, W = x.shape
    else:
        return False
    return N > 0 and C == w.shape[1] and _lib.lib().lsps_c8_conv3x3s2_ok(N, C, H, W, w.shape[0]) == 1


def c8_convT_s2_ok(x, w, stride, pad, outpad):
    if not _c8_enabled() or not options.get().c8s2 or stride != 2 or pad != 1 or outpad != 1 or \
            tuple(w.shape[2:]) != (3, 3):
        return False
    if is_c8(x):
        N, G, H, W, _ = x.shape
        C = G * 8
    elif x.dim() == 4:
        N, C, H, W = x.shape
    else:
        return False
    return N > 0 and C == w.shape[0] and _lib.lib().lsps_c8_convT3x3s2_ok(N, C, H, W, w.shape[1]) == 1


def _c8_act_backward(L, dy, y, slope, want_db, channels, st):
    """g = dy * LeakyReLU'(y) from the layer's OUTPUT (+ the bias gradient in the same pass) on C8 tensors."""
    if slope < 0:
        assert not want_db, "bias gradient of an activation-free C8 conv: not needed by any layer"
        return dy, None
    N, G, H, W, _ = dy.shape
    g = torch.empty_like(dy)
    db = torch.empty(channels, dtype=torch.float32, device=dy.device) if want_db else None
    ws, wsb = _lib.workspace(L.lsps_c8_act_bwd_bias_workspace_bytes(N, channels), dy.device)
    _lib.check(L.lsps_c8_act_bwd_bias(_lib.ptr(dy, BF16), _lib.ptr(y, BF16), _lib.ptr(g, BF16), _lib.ptr(db), N, channels, H * W,
                                      slope, ws, wsb, st), 'c8_act_bwd_bias')
    return g, db


class _ConvS2C8Fn(torch.autograd.Function):
    """LeakyReLUConv2d(C, K, 3, 2, 1) (common_net.py:246-256) on a C8 tensor: x [N][C/8][H][W][8] -> [N][K/8][H/2][W/2][8]."""

    @staticmethod
    def forward(ctx, x, w, b, slope, prev, own):
        L = _lib.lib()
        x, w = _c(x), _c(w)
        N, G, H, W, _ = x.shape
        C, K = G * 8, w.shape[0]
        ctx.prev, ctx.own = prev, own
        y = torch.empty((N, K // 8, H // 2, W // 2, 8), dtype=BF16, device=x.device)
        ws, wsb = _lib.workspace(L.lsps_c8_conv3x3s2_workspace_bytes(N, C, H, W, K), x.device)
        with profiler.span(2.0 * N * K * (H // 2) * (W // 2) * C * 9, 'c8s2_fwd_kernel'):
            _lib.check(L.lsps_c8_conv3x3s2_fwd(_lib.ptr(x, BF16), _lib.ptr(w), _lib.ptr(b), _lib.ptr(y, BF16), N, C, H, W, K, slope,
                                               ws, wsb, _lib.stream()), 'c8_conv3x3s2_fwd')
        ctx.geom = (N, C, H, W, K, slope)
        ctx.has_bias = b is not None
        ctx.save_for_backward(x, w, y if slope >= 0 else None)
        return y

    @staticmethod
    def backward(ctx, dy):
        L = _lib.lib()
        x, w, y = ctx.saved_tensors
        N, C, H, W, K, slope = ctx.geom
        dy = _c(dy)
        st = _lib.stream()
        flops = 2.0 * N * K * (H // 2) * (W // 2) * C * 9
        dx = dw = db = None
        want_db = ctx.has_bias and ctx.needs_input_grad[2]
        if ctx.own is not None and ctx.own.fused:        # the consumer's dgrad epilogue already applied this layer's LeakyReLU'
            g, db = dy, (ctx.own.db if want_db else None)
            ctx.own.fused, ctx.own.db = False, None
        else:
            g, db = _c8_act_backward(L, dy, y, slope, want_db, K, st)
        ws, wsb = _lib.workspace(L.lsps_c8_conv3x3s2_workspace_bytes(N, C, H, W, K), x.device)
        if ctx.needs_input_grad[1]:
            dw = torch.empty_like(w)
            with profiler.span(flops, 'c8s2_wgrad_kernel'):
                _lib.check(L.lsps_c8_conv3x3s2_wgrad(_lib.ptr(x, BF16), _lib.ptr(g, BF16), _lib.ptr(dw), N, C, H, W, K, ws, wsb, st),
                           'c8_conv3x3s2_wgrad')
        if ctx.needs_input_grad[0]:
            dx = torch.empty_like(x)
            with profiler.span(flops, 'c8s2_tr_kernel'):
                if _fusable(ctx.prev):                   # x is the previous layer's output: its LeakyReLU backward rides along
                    dbp = torch.empty(C, dtype=torch.float32, device=x.device)
                    _lib.check(L.lsps_c8_conv3x3s2_dgrad_act(_lib.ptr(g, BF16), _lib.ptr(w), _lib.ptr(x, BF16), ctx.prev.slope,
                                                             _lib.ptr(dx, BF16), _lib.ptr(dbp), N, C, H, W, K, ws, wsb, st),
                               'c8_conv3x3s2_dgrad_act')
                    ctx.prev.fused, ctx.prev.db = True, dbp
                else:
                    _lib.check(L.lsps_c8_conv3x3s2_dgrad(_lib.ptr(g, BF16), _lib.ptr(w), _lib.ptr(dx, BF16), N, C, H, W, K, ws, wsb, st),
                               'c8_conv3x3s2_dgrad')
        return dx, dw, db, None, None, None


def conv3x3s2_c8(x, w, b=None, slope=LRELU_SLOPE, prev=None, own=None):
    return _ConvS2C8Fn.apply(x, w, b, float(slope), prev, own)


class _ConvTS2C8Fn(torch.autograd.Function):
    """LeakyReLUConvTranspose2d(Ci, Co, 3, 2, 1, 1) (common_net.py:258-268) on a C8 tensor:
    x [N][Ci/8][H][W][8] -> [N][Co/8][2H][2W][8]."""

    @staticmethod
    def forward(ctx, x, w, b, slope, prev, own):
        L = _lib.lib()
        x, w = _c(x), _c(w)
        N, G, H, W, _ = x.shape
        Ci, Co = G * 8, w.shape[1]
        ctx.prev, ctx.own = prev, own
        y = torch.empty((N, Co // 8, 2 * H, 2 * W, 8), dtype=BF16, device=x.device)
        ws, wsb = _lib.workspace(L.lsps_c8_conv3x3s2_workspace_bytes(N, Co, 2 * H, 2 * W, Ci), x.device)
        with profiler.span(2.0 * N * Ci * H * W * Co * 9, 'c8s2_tr_kernel'):
            _lib.check(L.lsps_c8_convT3x3s2_fwd(_lib.ptr(x, BF16), _lib.ptr(w), _lib.ptr(b), _lib.ptr(y, BF16), N, Ci, H, W, Co, slope,
                                                ws, wsb, _lib.stream()), 'c8_convT3x3s2_fwd')
        ctx.geom = (N, Ci, H, W, Co, slope)
        ctx.has_bias = b is not None
        ctx.save_for_backward(x, w, y if slope >= 0 else None)
        return y

    @staticmethod
    def backward(ctx, dy):
        L = _lib.lib()
        x, w, y = ctx.saved_tensors
        N, Ci, H, W, Co, slope = ctx.geom
        dy = _c(dy)
        st = _lib.stream()
        flops = 2.0 * N * Ci * H * W * Co * 9
        dx = dw = db = None
        want_db = ctx.has_bias and ctx.needs_input_grad[2]
        if ctx.own is not None and ctx.own.fused:
            g, db = dy, (ctx.own.db if want_db else None)
            ctx.own.fused, ctx.own.db = False, None
        else:
            g, db = _c8_act_backward(L, dy, y, slope, want_db, Co, st)
        ws, wsb = _lib.workspace(L.lsps_c8_conv3x3s2_workspace_bytes(N, Co, 2 * H, 2 * W, Ci), x.device)
        if ctx.needs_input_grad[1]:
            dw = torch.empty_like(w)
            with profiler.span(flops, 'c8s2_wgrad_kernel'):
                _lib.check(L.lsps_c8_convT3x3s2_wgrad(_lib.ptr(x, BF16), _lib.ptr(g, BF16), _lib.ptr(dw), N, Ci, H, W, Co, ws, wsb, st),
                           'c8_convT3x3s2_wgrad')
        if ctx.needs_input_grad[0]:
            dx = torch.empty_like(x)
            with profiler.span(flops, 'c8s2_fwd_kernel'):
                if _fusable(ctx.prev):
                    dbp = torch.empty(Ci, dtype=torch.float32, device=x.device)
                    _lib.check(L.lsps_c8_convT3x3s2_dgrad_act(_lib.ptr(g, BF16), _lib.ptr(w), _lib.ptr(x, BF16), ctx.prev.slope,
                                                              _lib.ptr(dx, BF16), _lib.ptr(dbp), N, Ci, H, W, Co, ws, wsb, st),
                               'c8_convT3x3s2_dgrad_act')
                    ctx.prev.fused, ctx.prev.db = True, dbp
                else:
                    _lib.check(L.lsps_c8_convT3x3s2_dgrad(_lib.ptr(g, BF16), _lib.ptr(w), _lib.ptr(dx, BF16), N, Ci, H, W, Co, ws, wsb,
                                                          st), 'c8_convT3x3s2_dgrad')
        return dx, dw, db, None, None, None


def convT3x3s2_c8(x, w, b=None, slope=LRELU_SLOPE, prev=None, own=None):
    return _ConvTS2C8Fn.apply(x, w, b, float(slope), prev, own)


# ------------------------------------------------------------------------------------------
# f32 math mode: the 3x3 / stride-2 convs and transposed convs on the bf16 matrix pipe with THREE-LIMB operands ("X3",
# csrc/x3s2.h): an X3 tensor is an f32 activation carried as x = hi + mid + lo exactly (bf16 limbs), [N][3][C/8][H][W][8];
# six bf16 MFMAs per product, f32 accumulation: f32-class arithmetic (per-kernel error vs f64 at the level of the exact-f32
# kernels, tests/test_x3_gpu.py) at 0.5 - 0.7 of their time (profiles/r5b_x3s2_family_prototype.txt).
# ------------------------------------------------------------------------------------------
def is_x3(t):
    return torch.is_tensor(t) and t.dtype == BF16 and t.dim() == 6 and t.shape[1] == 3 and t.shape[-1] == 8


def x3_split(x):
    """f32 [N][C][H][W] -> X3 (no tape: the callers are autograd Functions)."""
    x = _c(x)
    N, C, H, W = x.shape
    y = torch.empty((N, 3, C // 8, H, W, 8), dtype=BF16, device=x.device)
    _lib.check(_lib.lib().lsps_x3_split_nchw(_lib.ptr(x), _lib.ptr(y, BF16), N, C, H * W, _lib.stream()), 'x3_split_nchw')
    return y


def x3_join(xl):
    xl = _c(xl)
    N, _, G, H, W, _ = xl.shape
    y = torch.empty((N, G * 8, H, W), dtype=torch.float32, device=xl.device)
    _lib.check(_lib.lib().lsps_x3_join_nchw(_lib.ptr(xl, BF16), _lib.ptr(y), N, G * 8, H * W, _lib.stream()), 'x3_join_nchw')
    return y


class _FromX3Fn(torch.autograd.Function):
    """X3 -> f32 NCHW (exact); backward: the f32 gradient split into limbs.  A safety net: run_layers only lets an X3 tensor out
    of a layer whose successor consumes X3."""

    @staticmethod
    def forward(ctx, xl):
        return x3_join(xl)

    @staticmethod
    def backward(ctx, g):
        return x3_split(g)


def _x3_shape(x):
    if is_x3(x):
        N, _, G, H, W, _ = x.shape
        return N, G * 8, H, W
    if torch.is_tensor(x) and x.dim() == 4 and x.dtype == torch.float32:
        return tuple(x.shape)
    return None


def _x3_enabled(macs):
    o = options.get()
    return get_math_mode() == 'f32' and o.x3 and macs >= o.x3_min_gmac * 1e9


def x3_conv_s2_ok(x, w, stride, pad):
    """LeakyReLUConv2d(C, K, 3, 2, 1) on `x` (f32 NCHW or X3) on the three-limb kernels?  f32 math mode, options.x3, the geometry the
    kernels cover (lsps_x3_conv3x3s2_ok) and at least options.x3_min_gmac x 10^9 multiply-adds (persistent 256-workgroup kernels)."""
    sh = _x3_shape(x)
    if sh is None or stride != 2 or pad != 1 or tuple(w.shape[2:]) != (3, 3):
        return False
    N, C, H, W = sh
    return N > 0 and C == w.shape[1] and _x3_enabled(9.0 * N * (H // 2) * (W // 2) * C * w.shape[0]) and \
        _lib.lib().lsps_x3_conv3x3s2_ok(N, C, H, W, w.shape[0]) == 1


def x3_convT_s2_ok(x, w, stride, pad, outpad):
    sh = _x3_shape(x)
    if sh is None or stride != 2 or pad != 1 or outpad != 1 or tuple(w.shape[2:]) != (3, 3):
        return False
    N, C, H, W = sh
    return N > 0 and C == w.shape[0] and _x3_enabled(9.0 * N * H * W * C * w.shape[1]) and _lib.lib().lsps_x3_convT3x3s2_ok(N, C, H, W, w.shape[1]) == 1


def _x3_grad_operand(L, dy, y, slope, own, want_db, channels, out_f32, st):
    """The X3 gradient w.r.t. a layer's PRE-activation output (what its wgrad / dgrad kernels read) and its bias gradient, from
    the gradient autograd hands the layer: f32 NCHW (the layer's output left the X3 family as f32: one pass applies LeakyReLU'(y)
    and splits) or X3 already multiplied by LeakyReLU' in the consumer's dgrad epilogue (`own.fused`)."""
    if own is not None and own.fused:
        db = own.db if want_db else None
        handed, own.grad = own.grad, None
        own.fused, own.db = False, None
        if handed is not None:
            return handed, db                            # X3, masked by an f32 consumer and handed over through the holder
        if not out_f32:
            return dy, db                                # X3, masked by the consumer
        slope, want_db_here = -1.0, False                # f32, masked by the consumer (1x1 head): split only
    else:
        if dy is None:
            raise _lib.LspsHipError("three-limb layer: no gradient arrived, neither along the autograd edge nor through the ActHolder")
        want_db_here = want_db
        db = None
        if not out_f32:                                  # an X3 output nobody masked (fusion switched off): via f32
            dy, y = x3_join(dy), (x3_join(y) if y is not None else None)
    dy = _c(dy)
    N, C, H, W = dy.shape
    g = torch.empty((N, 3, C // 8, H, W, 8), dtype=BF16, device=dy.device)
    dbh = torch.empty(channels, dtype=torch.float32, device=dy.device) if want_db_here else None
    ws, wsb = _lib.workspace(L.lsps_x3_act_bwd_bias_workspace_bytes(N, C), dy.device)
    _lib.check(L.lsps_x3_act_bwd_bias(_lib.ptr(dy), _lib.ptr(y) if slope >= 0 else None, _lib.ptr(g, BF16), _lib.ptr(dbh), N, C, H * W,
                                      slope, ws, wsb, st), 'x3_act_bwd_bias')
    return g, (dbh if want_db_here else db)


def x3_stem_ok(x, w, stride, pad):
    """7x7 one-input-channel stem writing its activation straight as X3 (f32 math mode; the consumer must be an X3 layer)."""
    if get_math_mode() != 'f32' or not options.get().x3 or not x3_fuse_enabled() or x.dim() != 4 or x.dtype != torch.float32:
        return False
    N, C, H, W = x.shape
    K, C2, R, S = w.shape
    return N > 0 and C == 1 and C2 == 1 and _lib.lib().lsps_x3_stem_ok(N, H, W, K, R, S, stride, pad) == 1


class _StemX3Fn(torch.autograd.Function):
    """LeakyReLUConv2d(1, 64, 7, stride, 3) (lsps_nets.py:117,184) with an X3 output.  The layer behind it (an X3 conv) applies this
    layer's LeakyReLU backward in its dgrad epilogue and hands the f32 NCHW result + the bias gradient back through `own`."""

    @staticmethod
    def forward(ctx, x, w, b, stride, pad, slope, own):
        L = _lib.lib()
        x, w = _c(x), _c(w)
        ctx.set_materialize_grads(False)
        N, _, H, W = x.shape
        K, _, R, S = w.shape
        P, Q = conv_out_size(H, R, stride, pad), conv_out_size(W, S, stride, pad)
        y = torch.empty((N, 3, K // 8, P, Q, 8), dtype=BF16, device=x.device)
        with profiler.span(2.0 * N * K * P * Q * R * S):
            _lib.check(L.lsps_x3_stem_fwd(_lib.ptr(x), _lib.ptr(w), _lib.ptr(b), _lib.ptr(y, BF16), N, H, W, K, R, S, stride, pad, slope,
                                          _lib.stream()), 'x3_stem_fwd')
        ctx.geom = (N, H, W, K, R, S, stride, pad)
        ctx.own, ctx.has_bias = own, b is not None
        ctx.save_for_backward(x, w)
        return y

    @staticmethod
    def backward(ctx, dy):
        L = _lib.lib()
        x, w = ctx.saved_tensors
        N, H, W, K, R, S, stride, pad = ctx.geom
        own = ctx.own
        if dy is not None or not own.fused or own.grad is None:
            raise _lib.LspsHipError("X3 stem: its output feeds exactly ONE three-limb conv, whose backward hands the gradient over "
                                    "through the ActHolder (got dy=%s, fused=%s)" % ('None' if dy is None else 'tensor', own.fused))
        g, db = own.grad, (own.db if (ctx.has_bias and ctx.needs_input_grad[2]) else None)
        own.fused, own.db, own.grad = False, None, None
        st = _lib.stream()
        flops = 2.0 * N * K * g.shape[2] * g.shape[3] * R * S
        ws, wsb = _lib.workspace(L.lsps_conv2d_workspace_bytes(N, 1, H, W, K, R, S, stride, pad), x.device)
        dx = dw = None
        if ctx.needs_input_grad[0]:
            dx = torch.empty_like(x)
            with profiler.span(flops):
                _lib.check(L.lsps_conv2d_dgrad(_lib.ptr(g), _lib.ptr(w), _lib.ptr(dx), N, 1, H, W, K, R, S, stride, pad, ws, wsb, st),
                           'conv2d_dgrad')
        if ctx.needs_input_grad[1]:
            dw = torch.empty_like(w)
            with profiler.span(flops):
                _lib.check(L.lsps_conv2d_wgrad(_lib.ptr(x), _lib.ptr(g), _lib.ptr(dw), None, N, 1, H, W, K, R, S, stride, pad, ws, wsb, st),
                           'conv2d_wgrad')
        return dx, dw, db, None, None, None, None


def stem_x3(x, w, b, stride, pad, slope, own):
    return _StemX3Fn.apply(x, w, b, int(stride), int(pad), float(slope), own)


class _ConvS2X3Fn(torch.autograd.Function):
    """LeakyReLUConv2d(C, K, 3, 2, 1) (common_net.py:246-256) on the three-limb kernels: x f32 NCHW (split here) or X3 (the
    output of the X3 layer in front: `prev`) -> y f32 NCHW (`out_f32`) or X3 (for the X3 layer behind: `own`)."""

    @staticmethod
    def forward(ctx, x, w, b, slope, prev, own, out_f32):
        L = _lib.lib()
        w = _c(w)
        ctx.set_materialize_grads(False)     # the gradient may arrive through the ActHolder instead of the autograd edge
        ctx.x_is_x3 = is_x3(x)
        xl = _c(x) if ctx.x_is_x3 else x3_split(x)
        N, _, G, H, W, _ = xl.shape
        C, K = G * 8, w.shape[0]
        ctx.prev, ctx.own, ctx.out_f32 = prev, own, bool(out_f32)
        P, Q = H // 2, W // 2
        y = torch.empty((N, K, P, Q), dtype=torch.float32, device=xl.device) if out_f32 else \
            torch.empty((N, 3, K // 8, P, Q, 8), dtype=BF16, device=xl.device)
        ws, wsb = _lib.workspace(L.lsps_x3_conv3x3s2_workspace_bytes(N, C, H, W, K), xl.device)
        with profiler.span(2.0 * N * K * P * Q * C * 9, 'x3s2_fwd_kernel'):
            _lib.check(L.lsps_x3_conv3x3s2_fwd(_lib.ptr(xl, BF16), _lib.ptr(w), _lib.ptr(b), _lib.ptr(y) if out_f32 else None,
                                               None if out_f32 else _lib.ptr(y, BF16), N, C, H, W, K, slope, ws, wsb, _lib.stream()),
                       'x3_conv3x3s2_fwd')
        ctx.geom = (N, C, H, W, K, slope)
        ctx.has_bias = b is not None
        ctx.save_for_backward(xl, w, y if slope >= 0 else None)
        return y

    @staticmethod
    def backward(ctx, dy):
        L = _lib.lib()
        xl, w, y = ctx.saved_tensors
        N, C, H, W, K, slope = ctx.geom
        st = _lib.stream()
        flops = 2.0 * N * K * (H // 2) * (W // 2) * C * 9
        dx = dw = None
        want_db = ctx.has_bias and ctx.needs_input_grad[2]
        g, db = _x3_grad_operand(L, dy, y, slope, ctx.own, want_db, K, ctx.out_f32, st)
        ws, wsb = _lib.workspace(L.lsps_x3_conv3x3s2_workspace_bytes(N, C, H, W, K), xl.device)
        if ctx.needs_input_grad[1]:
            dw = torch.empty_like(w)
            with profiler.span(flops, 'x3s2_wgrad_kernel'):
                _lib.check(L.lsps_x3_conv3x3s2_wgrad(_lib.ptr(xl, BF16), _lib.ptr(g, BF16), _lib.ptr(dw), N, C, H, W, K, ws, wsb, st),
                           'x3_conv3x3s2_wgrad')
        if ctx.needs_input_grad[0]:
            with profiler.span(flops, 'x3s2_tr_kernel'):
                if not ctx.x_is_x3:                      # the producer of x is an f32 layer: hand back f32 NCHW
                    dx = torch.empty((N, C, H, W), dtype=torch.float32, device=xl.device)
                    _lib.check(L.lsps_x3_conv3x3s2_dgrad(_lib.ptr(g, BF16), _lib.ptr(w), _lib.ptr(dx), None, None, 0.0, None, N, C, H, W, K,
                                                         ws, wsb, st), 'x3_conv3x3s2_dgrad')
                elif _fusable(ctx.prev) and ctx.prev.want == 'f32':
                    # x is the X3 output of an f32 layer (a 7x7 stem): its LeakyReLU backward rides along, the result goes back
                    # as f32 NCHW through the holder
                    gf = torch.empty((N, C, H, W), dtype=torch.float32, device=xl.device)
                    dbp = torch.empty(C, dtype=torch.float32, device=xl.device)
                    _lib.check(L.lsps_x3_conv3x3s2_dgrad(_lib.ptr(g, BF16), _lib.ptr(w), _lib.ptr(gf), None, _lib.ptr(xl, BF16),
                                                         ctx.prev.slope, _lib.ptr(dbp), N, C, H, W, K, ws, wsb, st),
                               'x3_conv3x3s2_dgrad(masked, f32)')
                    ctx.prev.fused, ctx.prev.db, ctx.prev.grad = True, dbp, gf
                else:
                    dx = torch.empty_like(xl)
                    if _fusable(ctx.prev):               # x is the previous X3 layer's output: its LeakyReLU backward rides along
                        dbp = torch.empty(C, dtype=torch.float32, device=xl.device)
                        _lib.check(L.lsps_x3_conv3x3s2_dgrad(_lib.ptr(g, BF16), _lib.ptr(w), None, _lib.ptr(dx, BF16), _lib.ptr(xl, BF16),
                                                             ctx.prev.slope, _lib.ptr(dbp), N, C, H, W, K, ws, wsb, st),
                                   'x3_conv3x3s2_dgrad(masked)')
                        ctx.prev.fused, ctx.prev.db = True, dbp
                    else:
                        _lib.check(L.lsps_x3_conv3x3s2_dgrad(_lib.ptr(g, BF16), _lib.ptr(w), None, _lib.ptr(dx, BF16), None, 0.0, None, N, C,
                                                             H, W, K, ws, wsb, st), 'x3_conv3x3s2_dgrad')
        return dx, dw, db, None, None, None, None


def conv3x3s2_x3(x, w, b=None, slope=LRELU_SLOPE, prev=None, own=None, out_f32=True):
    return _ConvS2X3Fn.apply(x, w, b, float(slope), prev, own, bool(out_f32))


class _ConvTS2X3Fn(torch.autograd.Function):
    """LeakyReLUConvTranspose2d(Ci, Co, 3, 2, 1, 1) (common_net.py:258-268) on the three-limb kernels: x [N,Ci,H,W] f32 (split
    here) or X3 -> y [N,Co,2H,2W] f32 NCHW (`out_f32`) or X3."""

    @staticmethod
    def forward(ctx, x, w, b, slope, prev, own, out_f32):
        L = _lib.lib()
        w = _c(w)
        ctx.set_materialize_grads(False)     # the gradient may arrive through the ActHolder instead of the autograd edge
        ctx.x_is_x3 = is_x3(x)
        xl = _c(x) if ctx.x_is_x3 else x3_split(x)
        N, _, G, H, W, _ = xl.shape
        Ci, Co = G * 8, w.shape[1]
        ctx.prev, ctx.own, ctx.out_f32 = prev, own, bool(out_f32)
        y = torch.empty((N, Co, 2 * H, 2 * W), dtype=torch.float32, device=xl.device) if out_f32 else \
            torch.empty((N, 3, Co // 8, 2 * H, 2 * W, 8), dtype=BF16, device=xl.device)
        ws, wsb = _lib.workspace(L.lsps_x3_conv3x3s2_workspace_bytes(N, Co, 2 * H, 2 * W, Ci), xl.device)
        with profiler.span(2.0 * N * Ci * H * W * Co * 9, 'x3s2_tr_kernel'):
            _lib.check(L.lsps_x3_convT3x3s2_fwd(_lib.ptr(xl, BF16), _lib.ptr(w), _lib.ptr(b), _lib.ptr(y) if out_f32 else None,
                                                None if out_f32 else _lib.ptr(y, BF16), N, Ci, H, W, Co, slope, ws, wsb, _lib.stream()),
                       'x3_convT3x3s2_fwd')
        ctx.geom = (N, Ci, H, W, Co, slope)
        ctx.has_bias = b is not None
        ctx.save_for_backward(xl, w, y if slope >= 0 else None)
        return y

    @staticmethod
    def backward(ctx, dy):
        L = _lib.lib()
        xl, w, y = ctx.saved_tensors
        N, Ci, H, W, Co, slope = ctx.geom
        st = _lib.stream()
        flops = 2.0 * N * Ci * H * W * Co * 9
        dx = dw = None
        want_db = ctx.has_bias and ctx.needs_input_grad[2]
        g, db = _x3_grad_operand(L, dy, y, slope, ctx.own, want_db, Co, ctx.out_f32, st)
        ws, wsb = _lib.workspace(L.lsps_x3_conv3x3s2_workspace_bytes(N, Co, 2 * H, 2 * W, Ci), xl.device)
        if ctx.needs_input_grad[1]:
            dw = torch.empty_like(w)
            with profiler.span(flops, 'x3s2_wgrad_kernel'):
                _lib.check(L.lsps_x3_convT3x3s2_wgrad(_lib.ptr(xl, BF16), _lib.ptr(g, BF16), _lib.ptr(dw), N, Ci, H, W, Co, ws, wsb, st),
                           'x3_convT3x3s2_wgrad')
        if ctx.needs_input_grad[0]:
            with profiler.span(flops, 'x3s2_fwd_kernel'):
                if not ctx.x_is_x3:
                    dx = torch.empty((N, Ci, H, W), dtype=torch.float32, device=xl.device)
                    _lib.check(L.lsps_x3_convT3x3s2_dgrad(_lib.ptr(g, BF16), _lib.ptr(w), _lib.ptr(dx), None, None, 0.0, None, N, Ci, H, W, Co,
                                                          ws, wsb, st), 'x3_convT3x3s2_dgrad')
                else:
                    dx = torch.empty_like(xl)
                    if _fusable(ctx.prev):               # x is the previous X3 layer's output: its LeakyReLU backward rides along
                        dbp = torch.empty(Ci, dtype=torch.float32, device=xl.device)
                        _lib.check(L.lsps_x3_convT3x3s2_dgrad(_lib.ptr(g, BF16), _lib.ptr(w), None, _lib.ptr(dx, BF16), _lib.ptr(xl, BF16),
                                                              ctx.prev.slope, _lib.ptr(dbp), N, Ci, H, W, Co, ws, wsb, st),
                                   'x3_convT3x3s2_dgrad(masked)')
                        ctx.prev.fused, ctx.prev.db = True, dbp
                    else:
                        _lib.check(L.lsps_x3_convT3x3s2_dgrad(_lib.ptr(g, BF16), _lib.ptr(w), None, _lib.ptr(dx, BF16), None, 0.0, None, N,
                                                              Ci, H, W, Co, ws, wsb, st), 'x3_convT3x3s2_dgrad')
        return dx, dw, db, None, None, None, None


def convT3x3s2_x3(x, w, b=None, slope=LRELU_SLOPE, prev=None, own=None, out_f32=True):
    return _ConvTS2X3Fn.apply(x, w, b, float(slope), prev, own, bool(out_f32))


def c8_stem_ok(x, w, stride, pad):
    """Single-input-channel stem (7x7) writing its activation straight in the C8 layout (bf16 math mode)."""
    if not _c8_enabled() or is_c8(x) or x.dim() != 4 or x.shape[1] != 1 or w.shape[1] != 1:
        return False
    N, _, H, W = x.shape
    return N > 0 and _lib.lib().lsps_c8_stem_ok(N, H, W, w.shape[0], w.shape[2], w.shape[3], stride, pad) == 1


class _StemC8Fn(torch.autograd.Function):
    """LeakyReLUConv2d(1, K, 7, stride, 3) (lsps_nets.py:117,184): f32 image in, C8 bf16 activation out.  Backward: weight and
    bias gradient in ONE kernel from (x, dy, saved y) — the LeakyReLU backward is applied while dy is staged."""

    @staticmethod
    def forward(ctx, x, w, b, stride, pad, slope):
        L = _lib.lib()
        x, w = _c(x), _c(w)
        N, _, H, W = x.shape
        K, _, R, S = w.shape
        P, Q = conv_out_size(H, R, stride, pad), conv_out_size(W, S, stride, pad)
        y = torch.empty((N, K // 8, P, Q, 8), dtype=BF16, device=x.device)
        with profiler.span(2.0 * N * K * P * Q * R * S):
            _lib.check(L.lsps_c8_stem_fwd(_lib.ptr(x), _lib.ptr(w), _lib.ptr(b), _lib.ptr(y, BF16), N, H, W, K, R, S, stride, pad, slope,
                                          _lib.stream()), 'c8_stem_fwd')
        ctx.geom = (N, H, W, K, R, S, stride, pad, slope, P, Q)
        ctx.has_bias = b is not None
        ctx.save_for_backward(x, w, y)
        return y

    @staticmethod
    def backward(ctx, dy):
        L = _lib.lib()
        x, w, y = ctx.saved_tensors
        N, H, W, K, R, S, stride, pad, slope, P, Q = ctx.geom
        dy = _c(dy)
        st = _lib.stream()
        flops = 2.0 * N * K * P * Q * R * S
        dx = dw = db = None
        want_db = ctx.has_bias and ctx.needs_input_grad[2]
        if ctx.needs_input_grad[1] or want_db:
            dw = torch.empty_like(w)
            db = torch.empty(K, dtype=torch.float32, device=x.device) if want_db else None
            ws, wsb = _lib.workspace(L.lsps_c8_stem_workspace_bytes(K, R, S), x.device)
            with profiler.span(flops):
                _lib.check(L.lsps_c8_stem_wgrad(_lib.ptr(x), _lib.ptr(dy, BF16), _lib.ptr(y, BF16), _lib.ptr(dw), _lib.ptr(db), N, H, W, K,
                                                R, S, stride, pad, slope if slope >= 0 else 1.0, ws, wsb, st), 'c8_stem_wgrad')
        if ctx.needs_input_grad[0] and slope >= 0 and L.lsps_c8_stem_dgrad_ok(N, H, W, K, R, S, stride, pad) == 1:
            # gradient w.r.t. the image (the discriminator's stems inside gen_update): bf16 tap GEMM + in-LDS col2im, one kernel
            dx = torch.empty_like(x)
            with profiler.span(flops):
                _lib.check(L.lsps_c8_stem_dgrad(_lib.ptr(dy, BF16), _lib.ptr(y, BF16), _lib.ptr(w), _lib.ptr(dx), N, H, W, K, R, S, stride,
                                                pad, slope, st), 'c8_stem_dgrad')
        elif ctx.needs_input_grad[0]:
            # shapes without that kernel: the f32 path of the layer
            g, _ = _c8_act_backward(L, dy, y, slope, False, K, st)
            g32 = torch.empty((N, K, P, Q), dtype=torch.float32, device=x.device)
            _lib.check(L.lsps_c8_to_nchw(_lib.ptr(g, BF16), _lib.ptr(g32), N, K, P * Q, st), 'c8_to_nchw')
            dx = torch.empty_like(x)
            ws, wsb = _lib.workspace(L.lsps_conv2d_workspace_bytes(N, 1, H, W, K, R, S, stride, pad), x.device)
            with profiler.span(flops):
                _lib.check(L.lsps_conv2d_dgrad(_lib.ptr(g32), _lib.ptr(w), _lib.ptr(dx), N, 1, H, W, K, R, S, stride, pad, ws, wsb, st),
                           'conv2d_dgrad')
        return dx, dw, db, None, None, None


def stem_c8(x, w, b, stride, pad, slope=LRELU_SLOPE):
    return _StemC8Fn.apply(x, w, b, int(stride), int(pad), float(slope))


def c8_pw1_ok(x, w, stride, pad, outpad):
    return _c8_enabled() and is_c8(x) and tuple(w.shape) == (x.shape[1] * 8, 1, 1, 1) and stride == 1 and pad == 0 and outpad == 0


class _Pw1C8Fn(torch.autograd.Function):
    """ConvTranspose2d(C, 1, kernel 1) [+ Tanh] on a C8 tensor (lsps_nets.py:226-229): x [N][C/8][H][W][8] -> y f32 [N,1,H,W]."""

    @staticmethod
    def forward(ctx, x, w, b, act, slope, prev):
        L = _lib.lib()
        x, w = _c(x), _c(w)
        N, G, H, W, _ = x.shape
        ctx.prev = prev
        y = torch.empty((N, 1, H, W), dtype=torch.float32, device=x.device)
        with profiler.span(2.0 * N * G * 8 * H * W, 'pw1_fwd_kernel'):
            _lib.check(L.lsps_c8_pw1_fwd(_lib.ptr(x, BF16), _lib.ptr(w), _lib.ptr(b), _lib.ptr(y), N, G * 8, H * W, act, slope,
                                         _lib.stream()), 'c8_pw1_fwd')
        ctx.geom = (N, G * 8, H, W, act, slope)
        ctx.has_bias = b is not None
        ctx.save_for_backward(x, w, y if act != ACT_NONE else None)
        return y

    @staticmethod
    def backward(ctx, dy):
        L = _lib.lib()
        x, w, y = ctx.saved_tensors
        N, C, H, W, act, slope = ctx.geom
        dy = _c(dy)
        st = _lib.stream()
        flops = 2.0 * N * C * H * W
        dx = dw = db = None
        if act != ACT_NONE:
            dpre = torch.empty_like(dy)
            _lib.check(L.lsps_act_bwd(_lib.ptr(dy), _lib.ptr(y), _lib.ptr(dpre), dy.numel(), act, slope, st), 'act_bwd')
        else:
            dpre = dy
        want_db = ctx.has_bias and ctx.needs_input_grad[2]
        want_w = ctx.needs_input_grad[1] or want_db
        if ctx.needs_input_grad[0]:
            dx = torch.empty_like(x)
            with profiler.span(flops, 'pw1_dgrad_kernel'):
                if _fusable(ctx.prev) and C <= 64:
                    # one pass over the head's input: its gradient with the previous layer's LeakyReLU backward and bias gradient,
                    # AND the head's own weight / bias gradient (same two operands)
                    dbp = torch.empty(C, dtype=torch.float32, device=x.device)
                    if want_w:
                        dw = torch.empty_like(w)
                        db = torch.empty(1, dtype=torch.float32, device=x.device) if want_db else None
                    wsd, wsdb = _lib.workspace(L.lsps_c8_pw1_dgrad_act_workspace_bytes(N, C), x.device)
                    _lib.check(L.lsps_c8_pw1_dgrad_act(_lib.ptr(dpre), _lib.ptr(w), _lib.ptr(x, BF16), ctx.prev.slope, _lib.ptr(dx, BF16),
                                                       _lib.ptr(dbp), _lib.ptr(dw), _lib.ptr(db), N, C, H * W, wsd, wsdb, st),
                               'c8_pw1_dgrad_act')
                    ctx.prev.fused, ctx.prev.db = True, dbp
                    want_w = False
                else:
                    _lib.check(L.lsps_c8_pw1_dgrad(_lib.ptr(dpre), _lib.ptr(w), _lib.ptr(dx, BF16), N, C, H * W, st), 'c8_pw1_dgrad')
        if want_w:
            dw = torch.empty_like(w)
            db = torch.empty(1, dtype=torch.float32, device=x.device) if want_db else None
            ws, wsb = _lib.workspace(L.lsps_c8_pw1_workspace_bytes(N, C), x.device)
            with profiler.span(flops, 'pw1_wgrad_kernel'):
                _lib.check(L.lsps_c8_pw1_wgrad(_lib.ptr(x, BF16), _lib.ptr(dpre), _lib.ptr(dw), _lib.ptr(db), N, C, H * W, ws, wsb, st),
                           'c8_pw1_wgrad')
        return dx, dw, db, None, None, None


def pw1_c8(x, w, b=None, act=ACT_NONE, slope=LRELU_SLOPE, prev=None):
    return _Pw1C8Fn.apply(x, w, b, int(act), float(slope), prev)


class _ResBlockC8Fn(torch.autograd.Function):
    """LeakyINSResBlock on C8 tensors as ONE autograd node (common_net.py:160-181): both convs run c8_conv3x3_kernel with
    the InstanceNorm (+ LeakyReLU | + skip) in the epilogue; backward = norm-2 backward, two transposing-read weight
    gradients, conv-2 dgrad through norm-1 + LeakyReLU backward in its epilogue, conv-1 dgrad + skip gradient.  Saved for
    backward: x, a1, y (bf16) and the two rstd vectors."""

    @staticmethod
    def forward(ctx, x, w1, w2):
        L = _lib.lib()
        x, w1, w2 = _c(x), _c(w1), _c(w2)
        N, G, H, W, _ = x.shape
        C = G * 8
        K = w1.shape[0]
        assert w1.shape == (K, C, 3, 3) and w2.shape == (K, K, 3, 3) and K == C, "residual block: C -> C, 3x3"
        st = _lib.stream()
        ws, wsb = _lib.workspace(L.lsps_c8_conv3x3_workspace_bytes(C, K), x.device)
        flops = 2.0 * N * K * H * W * C * 9
        a1 = torch.empty_like(x)
        y = torch.empty_like(x)
        r1 = torch.empty(N * K, dtype=torch.float32, device=x.device)
        r2 = torch.empty_like(r1)
        with profiler.span(flops, 'c8_conv3x3_kernel'):
            _lib.check(L.lsps_c8_conv3x3_in_fwd(_lib.ptr(x, BF16), _lib.ptr(w1), None, _lib.ptr(a1, BF16), _lib.ptr(r1), N, C, H, W, K,
                                                LRELU_SLOPE, IN_EPS, ws, wsb, st), 'c8_conv3x3_in_fwd')
        with profiler.span(flops, 'c8_conv3x3_kernel'):
            _lib.check(L.lsps_c8_conv3x3_in_fwd(_lib.ptr(a1, BF16), _lib.ptr(w2), _lib.ptr(x, BF16), _lib.ptr(y, BF16), _lib.ptr(r2), N, K,
                                                H, W, K, -1.0, IN_EPS, ws, wsb, st), 'c8_conv3x3_in_fwd')
        ctx.save_for_backward(x, w1, w2, a1, y, r1, r2)
        return y

    @staticmethod
    def backward(ctx, g):
        L = _lib.lib()
        x, w1, w2, a1, y, r1, r2 = ctx.saved_tensors
        g = _c(g)
        N, G, H, W, _ = x.shape
        C = K = G * 8
        st = _lib.stream()
        flops = 2.0 * N * K * H * W * C * 9
        ws, wsb = _lib.workspace(L.lsps_c8_conv3x3_workspace_bytes(C, K), x.device)
        dh2 = torch.empty_like(y)
        _lib.check(L.lsps_c8_inorm_bwd(_lib.ptr(g, BF16), _lib.ptr(y, BF16), _lib.ptr(x, BF16), _lib.ptr(r2), _lib.ptr(dh2, BF16), N, K,
                                       H * W, -1.0, st), 'c8_inorm_bwd')
        dw1 = dw2 = dx = None

        def wgrad(inp, dy):
            dw = torch.empty((K, C, 3, 3), dtype=torch.float32, device=x.device)
            wsw, wswb = _lib.workspace(L.lsps_c8_conv3x3_wgrad_workspace_bytes(N, C, K), x.device)
            with profiler.span(flops, 'c8_wgrad_kernel'):
                _lib.check(L.lsps_c8_conv3x3_wgrad(_lib.ptr(inp, BF16), _lib.ptr(dy, BF16), _lib.ptr(dw), N, C, H, W, K, wsw, wswb, st),
                           'c8_conv3x3_wgrad')
            return dw
        if ctx.needs_input_grad[2]:
            dw2 = wgrad(a1, dh2)
        dh1 = torch.empty_like(a1)
        # the weight-gradient partial sums and the packed weights share the per-stream workspace: the pack happens inside the
        # dgrad call below, after the weight-gradient launches that read the partials were enqueued on the same stream
        ws, wsb = _lib.workspace(L.lsps_c8_conv3x3_workspace_bytes(C, K), x.device)
        with profiler.span(flops, 'c8_conv3x3_kernel'):
            _lib.check(L.lsps_c8_conv3x3_dgrad_inbwd(_lib.ptr(dh2, BF16), _lib.ptr(w2), _lib.ptr(a1, BF16), _lib.ptr(r1), _lib.ptr(dh1, BF16),
                                                     N, K, H, W, K, LRELU_SLOPE, ws, wsb, st), 'c8_conv3x3_dgrad_inbwd')
        if ctx.needs_input_grad[1]:
            dw1 = wgrad(x, dh1)
        if ctx.needs_input_grad[0]:
            dx = dh2                                     # dh2 is dead: its storage receives dx
            ws, wsb = _lib.workspace(L.lsps_c8_conv3x3_workspace_bytes(C, K), x.device)
            with profiler.span(flops, 'c8_conv3x3_kernel'):
                _lib.check(L.lsps_c8_conv3x3_dgrad_acc(_lib.ptr(dh1, BF16), _lib.ptr(w1), _lib.ptr(g, BF16), _lib.ptr(dx, BF16), N, C, H, W,
                                                       K, ws, wsb, st), 'c8_conv3x3_dgrad_acc')
        return dx, dw1, dw2


def res_block_c8(x, w1, w2):
    return _ResBlockC8Fn.apply(x, w1, w2)


def res_block(x, w1, w2):
    """x + IN(conv3x3(LReLU(IN(conv3x3(x, w1))), w2)) — LeakyINSResBlock (common_net.py:160-181), one autograd node."""
    if x.shape[0] == 0:
        return _empty(x, x.shape, w1, w2)
    return _ResBlockFn.apply(x, w1, w2, not torch.is_grad_enabled())


def instance_norm_(y, residual=None, slope=-1.0):
    """In-place fused InstanceNorm: y <- act(IN(y)) (+ residual).  slope < 0: no activation."""
    if y.numel() == 0:
        return y
    return _InormFn.apply(y, residual, float(slope))


# ------------------------------------------------------------------------------------------
# losses
# ------------------------------------------------------------------------------------------
class _LossFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, kind, denom, a, b):
        L = _lib.lib()
        a = _c(a)
        b = _c(b) if b is not None else None
        if b is not None:
            assert a.shape == b.shape, "loss operands must have equal shapes"
        out = torch.empty(1, dtype=torch.float32, device=a.device)
        n = a.numel()
        ws, wsb = _lib.workspace(L.lsps_loss_workspace_bytes(n), a.device)
        _lib.check(L.lsps_loss_fwd(kind, _lib.ptr(a), _lib.ptr(b), n, denom, _lib.ptr(out), ws, wsb, _lib.stream()),
                   'loss_fwd')
        ctx.kind, ctx.denom = kind, denom
        ctx.save_for_backward(a, b)
        return out.reshape(())

    @staticmethod
    def backward(ctx, g):
        L = _lib.lib()
        a, b = ctx.saved_tensors
        g = _c(g.reshape(1))
        da = torch.empty_like(a)
        need_b = b is not None and ctx.needs_input_grad[3]
        db = torch.empty_like(b) if need_b else None
        _lib.check(L.lsps_loss_bwd(ctx.kind, _lib.ptr(a), _lib.ptr(b), a.numel(), ctx.denom, _lib.ptr(g),
                                   _lib.ptr(da), _lib.ptr(db), _lib.stream()), 'loss_bwd')
        return None, None, da, db


def l1_loss(a, b=None):
    """nn.L1Loss()(a, b) (mean); b=None is the feature-matching form against zeros (lsps_trainer.py:44-49,172-177)."""
    return _LossFn.apply(LOSS_L1, float(a.numel()), a, b)


def l2_loss(a, b):
    """torch.pow(a-b, 2).mean() (lsps_trainer.py:51-52)."""
    return _LossFn.apply(LOSS_L2, float(a.numel()), a, b)


def kl_loss(mu, sd=None):
    """_compute_kl (lsps_trainer.py:55-60): mean(mu^2), or sum(mu^2+sd^2-log sd^2)/batch."""
    if sd is None:
        return _LossFn.apply(LOSS_SQ, float(mu.numel()), mu, None)
    return _LossFn.apply(LOSS_KLSD, float(mu.size(0)), mu, sd)


class _BceFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, logits, target):
        L = _lib.lib()
        logits = _c(logits)
        n = logits.numel()
        out = torch.empty(3, dtype=torch.float32, device=logits.device)
        ws, wsb = _lib.workspace(L.lsps_loss_workspace_bytes(n), logits.device)
        _lib.check(L.lsps_bce_sigmoid_fwd(_lib.ptr(logits), n, target, _lib.ptr(out), ws, wsb, _lib.stream()), 'bce_fwd')
        ctx.target = target
        ctx.save_for_backward(logits)
        counts = out[1:3]
        ctx.mark_non_differentiable(counts)
        return out[0], counts

    @staticmethod
    def backward(ctx, g, _gc):
        L = _lib.lib()
        (logits,) = ctx.saved_tensors
        g = _c(g.reshape(1))
        dx = torch.empty_like(logits)
        _lib.check(L.lsps_bce_sigmoid_bwd(_lib.ptr(logits), logits.numel(), ctx.target, _lib.ptr(g), _lib.ptr(dx),
                                          _lib.stream()), 'bce_bwd')
        return dx, None


def bce_sigmoid(logits, target):
    """binary_cross_entropy(sigmoid(logits), const target) (lsps_trainer.py:107-112,179-192).
    Returns (loss, counts) with counts = [#(p>=0.5), #(p<=0.5)] for helpers.py:20-32."""
    return _BceFn.apply(logits, float(target))


# ------------------------------------------------------------------------------------------
# pose-MLP linear
# ------------------------------------------------------------------------------------------
class _LinearFn(torch.autograd.Function):
    """nn.Linear (+ LeakyReLU | Softplus) — lsps_nets.py:44-50, 73-83; common_net.py:221-231."""

    @staticmethod
    def forward(ctx, x, w, b, act, slope):
        L = _lib.lib()
        x, w = _c(x), _c(w)
        n, i = x.shape
        o = w.shape[0]
        y = torch.empty((n, o), dtype=torch.float32, device=x.device)
        _lib.check(L.lsps_linear_fwd(_lib.ptr(x), _lib.ptr(w), _lib.ptr(b), _lib.ptr(y), n, i, o, act, slope,
                                     _lib.stream()), 'linear_fwd')
        ctx.act, ctx.slope = act, slope
        ctx.save_for_backward(x, w, y)
        return y

    @staticmethod
    def backward(ctx, dy):
        L = _lib.lib()
        x, w, y = ctx.saved_tensors
        dy = _c(dy)
        n, i = x.shape
        o = w.shape[0]
        dx = torch.empty_like(x) if ctx.needs_input_grad[0] else None
        dw = torch.empty_like(w)
        db = torch.empty(o, dtype=torch.float32, device=x.device)
        dz = torch.empty_like(y)
        _lib.check(L.lsps_linear_bwd(_lib.ptr(x), _lib.ptr(w), _lib.ptr(y), _lib.ptr(dy), _lib.ptr(dx), _lib.ptr(dw),
                                     _lib.ptr(db), n, i, o, ctx.act, ctx.slope, _lib.ptr(dz), _lib.stream()), 'linear_bwd')
        return dx, dw, db, None, None


def linear(x, w, b, act=ACT_NONE, slope=LRELU_SLOPE):
    if x.shape[0] == 0:
        return _empty(x, (0, w.shape[0]), w, b)
    return _LinearFn.apply(x, w, b, int(act), float(slope))


# ------------------------------------------------------------------------------------------
# elementwise glue
# ------------------------------------------------------------------------------------------
class _AxpyFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, y, alpha):
        L = _lib.lib()
        x, y = _c(x), _c(y)
        assert x.shape == y.shape
        out = torch.empty_like(x)
        _lib.check(L.lsps_axpy(_lib.ptr(x), _lib.ptr(y), alpha, _lib.ptr(out), x.numel(), _lib.stream()), 'axpy')
        ctx.alpha = alpha
        return out

    @staticmethod
    def backward(ctx, g):
        gy = None
        if ctx.needs_input_grad[1]:
            gy = g * ctx.alpha
        return g, gy, None


class _MulAddFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, t, m):
        L = _lib.lib()
        x, t, m = _c(x), _c(t), _c(m)
        assert x.shape == t.shape == m.shape
        out = torch.empty_like(t)
        _lib.check(L.lsps_mul_add(_lib.ptr(x), _lib.ptr(t), _lib.ptr(m), _lib.ptr(out), t.numel(), _lib.stream()), 'mul_add')
        ctx.save_for_backward(m)
        return out

    @staticmethod
    def backward(ctx, g):
        (m,) = ctx.saved_tensors
        g = _c(g)
        gt = None
        if ctx.needs_input_grad[1]:
            gt = torch.empty_like(g)
            _lib.check(_lib.lib().lsps_mul_add(None, _lib.ptr(g), _lib.ptr(m), _lib.ptr(gt), g.numel(), _lib.stream()),
                       'mul_add')
        return g, gt, None


def mul_add(x, t, m):
    """x + t*m: dropout mask (already divided by 1-p) on a residual branch, then the skip connection
    (common_net.py:171-172,180)."""
    if x.numel() == 0:
        return x
    return _MulAddFn.apply(x, t, m)


class _BnormFn(torch.autograd.Function):
    """nn.BatchNorm2d / nn.BatchNorm1d [+ nn.LeakyReLU] of the BN block variants (common_net.py:183-322).  `x`: [N, C, ...]."""

    @staticmethod
    def forward(ctx, x, gamma, beta, run_mean, run_var, training, slope, eps, momentum):
        L = _lib.lib()
        x = _c(x)
        N, C = x.shape[0], x.shape[1]
        HW = x.numel() // (N * C)
        y = torch.empty_like(x)
        mean = torch.empty(C, dtype=torch.float32, device=x.device)
        rstd = torch.empty_like(mean)
        ws, wsb = _lib.workspace(L.lsps_bnorm_workspace_bytes(C), x.device)
        _lib.check(L.lsps_bnorm_fwd(_lib.ptr(x), _lib.ptr(gamma), _lib.ptr(beta), _lib.ptr(run_mean), _lib.ptr(run_var),
                                    _lib.ptr(y), _lib.ptr(mean), _lib.ptr(rstd), N, C, HW, int(bool(training)), eps, momentum,
                                    slope, ws, wsb, _lib.stream()), 'bnorm_fwd')
        ctx.geom = (N, C, HW, bool(training), slope)
        ctx.save_for_backward(x, gamma, mean, rstd, y if slope >= 0 else None)
        return y

    @staticmethod
    def backward(ctx, g):
        L = _lib.lib()
        x, gamma, mean, rstd, y = ctx.saved_tensors
        N, C, HW, training, slope = ctx.geom
        g = _c(g)
        st = _lib.stream()
        if slope >= 0:
            gp = torch.empty_like(g)
            _lib.check(L.lsps_act_bwd(_lib.ptr(g), _lib.ptr(y), _lib.ptr(gp), g.numel(), ACT_LRELU, slope, st), 'act_bwd')
            g = gp
        dx = torch.empty_like(x)
        dgamma = torch.empty(C, dtype=torch.float32, device=x.device) if ctx.needs_input_grad[1] else None
        dbeta = torch.empty(C, dtype=torch.float32, device=x.device) if ctx.needs_input_grad[2] else None
        ws, wsb = _lib.workspace(L.lsps_bnorm_workspace_bytes(C), x.device)
        _lib.check(L.lsps_bnorm_bwd(_lib.ptr(g), _lib.ptr(x), _lib.ptr(mean), _lib.ptr(rstd), _lib.ptr(gamma), _lib.ptr(dx),
                                    _lib.ptr(dgamma), _lib.ptr(dbeta), N, C, HW, int(training), ws, wsb, st), 'bnorm_bwd')
        return dx, dgamma, dbeta, None, None, None, None, None, None


def batch_norm(x, gamma=None, beta=None, run_mean=None, run_var=None, training=True, slope=-1.0, eps=1e-5, momentum=0.1):
    """act(BN(x)); running statistics are updated in place in training mode.  slope < 0: no activation."""
    if x.shape[0] == 0:
        return _empty(x, x.shape, gamma, beta)
    return _BnormFn.apply(x, gamma, beta, run_mean, run_var, bool(training), float(slope), float(eps), float(momentum))


class _ActFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, kind, slope):
        L = _lib.lib()
        x = _c(x)
        out = torch.empty_like(x)
        _lib.check(L.lsps_act_fwd(_lib.ptr(x), _lib.ptr(out), x.numel(), kind, slope, _lib.stream()), 'act_fwd')
        ctx.kind, ctx.slope = kind, slope
        ctx.save_for_backward(out)
        return out

    @staticmethod
    def backward(ctx, g):
        (out,) = ctx.saved_tensors
        g = _c(g)
        dx = torch.empty_like(g)
        _lib.check(_lib.lib().lsps_act_bwd(_lib.ptr(g), _lib.ptr(out), _lib.ptr(dx), g.numel(), ctx.kind, ctx.slope,
                                           _lib.stream()), 'act_bwd')
        return dx, None, None


def act(x, kind, slope=0.0):
    """Standalone activation: nn.ReLU (kind=ACT_LRELU, slope=0; common_net.py:146,361), nn.Softplus, nn.Tanh."""
    if x.numel() == 0:
        return x
    return _ActFn.apply(x, int(kind), float(slope))


# ------------------------------------------------------------------------------------------
# scoped cache of packed weight panels (lsps_pack_cache_begin / _end)
# ------------------------------------------------------------------------------------------
_pack_arenas = {}
PACK_CACHE_BYTES = 1 << 30          # gen + dis panels in both directions are ~0.4 GB


def weight_cache_begin(device):
    """From here until `weight_cache_end()` the conv weights are promised not to change (one update method of the
    trainer up to its optimizer step): packed weight panels are built once and reused."""
    if not options.get().pack_cache:
        return
    key = (device.type, device.index)
    buf = _pack_arenas.get(key)
    if buf is None:
        buf = _pack_arenas[key] = torch.empty(PACK_CACHE_BYTES, dtype=torch.uint8, device=device)
    _lib.check(_lib.lib().lsps_pack_cache_begin(buf.data_ptr(), buf.numel()), 'pack_cache_begin')


def weight_cache_end():
    _lib.check(_lib.lib().lsps_pack_cache_end(), 'pack_cache_end')


FROZEN_CACHE_MAX_BYTES = 1 << 29    # the generator's panels for the forward direction: 28 x 9.4 MB of F(4x4,3x3) U + small ones
_frozen_state = {'resets': 0, 'last': None}


def frozen_resets():
    """How often the library's (process-wide) frozen table has been re-targeted: another arena, another buffer or another epoch
    (each empties the table).  A hipGraph that was captured while panels were frozen is only valid under the count it saw."""
    return _frozen_state['resets']


def weight_cache_frozen(arena=None, epoch=0):
    """Declares the weights inside `arena` (an optim.FlatArena) unchanged ACROSS the scopes that follow, until `epoch` changes
    (lsps_pack_cache_frozen): their packed panels survive `weight_cache_end()`.  None: no frozen weights.
    The panels live in a buffer that belongs to the ARENA (`arena._frozen_buf`: sized from its parameter bytes — the F(4x4,3x3)
    transform is 4x a 3x3 weight, the other panels ~1.1x —, capped at 512 MiB, freed with the trainer): another trainer's
    declaration re-targets the library's table but can never overwrite panels a captured graph of this trainer reads."""
    L = _lib.lib()
    if arena is None or not options.get().pack_cache or not options.get().frozen_packs:
        _lib.check(L.lsps_pack_cache_frozen(None, None, None, 0, 0), 'pack_cache_frozen')
        return False
    flat_params = arena.flat_p
    buf = getattr(arena, '_frozen_buf', None)
    if buf is None:
        nbytes = min(FROZEN_CACHE_MAX_BYTES, max(1 << 20, 6 * flat_params.numel() * flat_params.element_size()))
        buf = arena._frozen_buf = torch.empty((nbytes + 255) // 256 * 256, dtype=torch.uint8, device=flat_params.device)
    lo = flat_params.data_ptr()
    target = (lo, flat_params.numel(), buf.data_ptr(), buf.numel(), int(epoch) & 0xffffffffffffffff)
    if target != _frozen_state['last']:             # the same comparison the library makes before it empties its table
        _frozen_state['last'] = target
        _frozen_state['resets'] += 1
    _lib.check(L.lsps_pack_cache_frozen(lo, lo + flat_params.numel() * flat_params.element_size(), buf.data_ptr(), buf.numel(),
                                        target[4]), 'pack_cache_frozen')
    return True


def axpy(x, y, alpha=1.0):
    """x + alpha*y (GaussianNoiseLayer: common_net.py:39-40; reparameterisation: lsps_nets.py:78)."""
    if x.numel() == 0:
        return x
    return _AxpyFn.apply(x, y, float(alpha))
