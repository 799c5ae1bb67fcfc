"""LSPSTrainer on the HIP kernels (reference: src/trainers/lsps_trainer.py:16-350).

Same constructor, attributes (`dis gen vae map`, `*_opt`, `*_sch`, loss / accuracy scalars as
numpy values) and method signatures as the reference, so `depth_train.py`'s loop drives it
unchanged.  Differences, none of which changes a result:
  * back-propagation the reference computes and then throws away is not launched: `dis_update`
    and `post_update` run the generator without a tape (the reference back-props into `gen` and
    zeroes/ignores those grads, :77,:144,:221), `gen_update` freezes the discriminator weights
    (its dis grads are zeroed at :144);
  * the optimizers step a flat HBM arena with one fused Adam launch (lsps_amd/optim.py);
  * under `torch.distributed` (one process per GPU) gradients are all-reduced over RCCL in a few
    large buckets overlapped with backward (lsps_amd/dist.py); single-process behaviour is unchanged;
  * all per-step scalars are fetched with ONE device->host copy instead of one per loss;
  * keyword-only `noise=` arguments inject the random draws for parity tests.
"""
import os
import pickle as _pickle

import numpy as np
import torch
import torch.nn as nn

from .lsps_nets import *  # noqa: F401,F403
from .lsps_nets import Mapping, SharedDis, SharedResGen, SharedResXGen, poseVAE
from .helpers import get_model_list, _compute_fake_acc, _compute_true_acc  # noqa: F401
from .init import *  # noqa: F401,F403
from .init import gaussian_weights_init
from .. import dist as lsps_dist
from .. import ops
from ..optim import FlatAdam

_NETS = {'Mapping': Mapping, 'poseVAE': poseVAE, 'SharedDis': SharedDis, 'SharedResGen': SharedResGen,
         'SharedResXGen': SharedResXGen}


def _net(spec):
    name = spec['name']
    if name not in _NETS:
        raise KeyError("unknown network '%s' (available: %s)" % (name, sorted(_NETS)))
    return _NETS[name](spec)


def _weights_constant(fn):
    """The conv weights do not change inside an update method until its optimizer step (`_step` ends the scope just
    before it): packed weight panels are cached for that long (ops.weight_cache_begin).  With `use_graphs(True)` the
    method's launches (forward, losses, backward: everything up to the optimizer step) are captured into a hipGraph the
    second time a call signature is seen and replayed from then on (`_GraphedUpdate`)."""
    import functools

    def eager(self, images_a, *a, **k):
        self._declare_frozen(fn.__name__)
        ops.weight_cache_begin(images_a.device)
        try:
            return fn(self, images_a, *a, **k)
        finally:
            ops.weight_cache_end()
            ops.weight_cache_frozen(None)   # the promise ends with the method: scopes opened elsewhere see no frozen weights

    def with_modes(self, *a, **k):
        # The library's math / Winograd modes are process-wide (include/lsps_hip.h); a trainer that was given its own
        # (`set_modes`) installs them for the duration of its update methods, so two trainers in different modes can
        # alternate in one process (one thread drives one device: no concurrent update methods).
        if self.math_mode is None and self.winograd is None:
            return dispatch(self, *a, **k)
        prev = (ops.get_math_mode(), ops.get_winograd())
        try:
            if self.math_mode is not None:
                ops.set_math_mode(self.math_mode)
            if self.winograd is not None:
                ops.set_winograd(self.winograd)
            return dispatch(self, *a, **k)
        finally:
            ops.set_math_mode(prev[0])
            ops.set_winograd(prev[1])

    def dispatch(self, images_a, *a, **k):
        if not self._graphs_on or (lsps_dist.active() and not lsps_dist.capturable()):
            return eager(self, images_a, *a, **k)
        return self._graphed(fn.__name__, eager, (images_a,) + a, k)
    return functools.wraps(fn)(with_modes)


def _flatten_tensors(obj, out):
    """Tensor leaves of nested tuples / lists / dicts in a fixed order; returns a hashable description of the rest."""
    if torch.is_tensor(obj):
        out.append(obj)
        return ('T', tuple(obj.shape), str(obj.dtype))
    if isinstance(obj, (tuple, list)):
        return (type(obj).__name__,) + tuple(_flatten_tensors(o, out) for o in obj)
    if isinstance(obj, dict):
        return ('dict',) + tuple((k, _flatten_tensors(obj[k], out)) for k in sorted(obj))
    if isinstance(obj, np.ndarray) and obj.size > 1:
        # repr() of a large array elides its middle: two different arrays would share one signature and the captured
        # copy would go stale
        raise TypeError("hipGraph replay: pass arrays as tensors (got a numpy array of %d elements)" % obj.size)
    return ('V', repr(obj))


def _rebuild(obj, it):
    if torch.is_tensor(obj):
        return next(it)
    if isinstance(obj, (tuple, list)):
        return type(obj)(_rebuild(o, it) for o in obj)
    if isinstance(obj, dict):
        return dict((k, _rebuild(obj[k], it)) for k in sorted(obj))
    return obj


class _GraphedUpdate(object):
    """One update method at one call signature as a hipGraph: static copies of the tensor arguments, the captured launches
    (zero_grad memset, forward, losses, backward into the gradient arena, the stacked loss scalars), and the host-side
    facts a replay must restore (which parameters got a gradient; the pending optimizer step)."""

    def __init__(self, trainer, eager, args, kwargs, pool):
        self.static_in = []
        _flatten_tensors((args, kwargs), self.static_in)
        self.static_in = [t.clone() for t in self.static_in]
        a, k = _rebuild((args, kwargs), iter(self.static_in))
        self.graph = torch.cuda.CUDAGraph()
        trainer._capturing = self
        self.pending = None
        # Under data parallelism the reducer's hooks fire during the capture like in any eager step, so the bucket
        # all-reduces (RCCL kernels on the process group's stream, forked from / joined to the launch stream by events)
        # become nodes of the graph at the points of backward where they are launched.  RCCL's watchdog thread polls
        # events meanwhile: legal only if the capture does not police other threads ('thread_local').
        mode = dict(capture_error_mode='thread_local') if lsps_dist.active() else {}
        try:
            with torch.cuda.graph(self.graph, pool=pool, **mode):
                self.result = eager(trainer, *a, **k)
        finally:
            trainer._capturing = None
        assert self.pending is not None, "update method did not reach _step"
        key, opt, names, scal = self.pending
        self.touched = list(opt.arena.touched)

    def replay(self, trainer, args, kwargs):
        fresh = []
        _flatten_tensors((args, kwargs), fresh)
        for dst, src in zip(self.static_in, fresh):
            if dst.data_ptr() != src.data_ptr():
                dst.copy_(src)
        self.graph.replay()
        key, opt, names, scal = self.pending
        opt.arena.touched[:] = self.touched
        # data parallel: the captured all-reduce summed `scal` over the ranks in place
        trainer._finish_step(opt, names, scal, scal / float(lsps_dist.world()) if lsps_dist.active() else None)
        return self.result


# The scalars the update methods publish (reference: lsps_trainer.py:73,132-140,198-199,214-217,260-261).  The reference copies each
# to the host synchronously (`.data.cpu().numpy()`); here ONE asynchronous device -> pinned-host copy per step carries all of them and
# the attribute materialises as the same numpy value when it is READ (class-level properties below): a step no longer ends with a
# host synchronisation, so the host queues the next update method while the GPU still runs this one (round 6: the per-launch trace of the
# bs = 128 step showed ~1.1 ms of idle GPU after each of the two synchronising copies; the 5 ms estimate3 step pays the same per step).
_SCALAR_NAMES = ('gen_enc_loss', 'gen_enc_loss2', 'gen_ad_loss', 'gen_ll_loss', 'gen_ll_loss2', 'gen_map_loss', 'gen_map_loss2',
                'gen_total_loss', 'dis_ad_loss', 'dis_feat_loss', 'dis_loss', 'dis_true_acc', 'dis_fake_acc', 'dis_reg_loss',
                'dis_total_loss', 'vae_total_loss')


class _ScalarSlot(object):
    """One pinned host buffer of a small ring + the event behind its copy + the attribute names that currently live in it."""
    __slots__ = ('host', 'event', 'names')

    def __init__(self):
        self.host = torch.empty(32, dtype=torch.float32).pin_memory()
        self.event = torch.cuda.Event()
        self.names = []


def _scalar_property(name):
    def get(self):
        d = self.__dict__
        v = d.get(name, _UNSET)
        if v is _UNSET:
            raise AttributeError(name)
        if isinstance(v, tuple) and len(v) == 2 and isinstance(v[0], _ScalarSlot):      # still in flight: wait for ITS copy only
            slot, i = v
            slot.event.synchronize()
            v = d[name] = np.asarray(slot.host[i].item(), dtype=np.float32)
        return v

    def set(self, value):
        self.__dict__[name] = value

    def delete(self):
        del self.__dict__[name]
    return property(get, set, delete)


_UNSET = object()


class LSPSTrainer(nn.Module):
    def __init__(self, hyperparameters):
        super(LSPSTrainer, self).__init__()
        ops.options.warn_if_env_changed()     # LSPS_* set after `import lsps_amd` are NOT in force until options.reload_env()
        self.lazy_scalars = ops.options.get().lazy_scalars    # loss / accuracy scalars materialise when read (False: synchronous copy per step)
        lr = hyperparameters['lr']
        self.dis = _net(hyperparameters['dis'])
        self.gen = _net(hyperparameters['gen'])
        self.vae = _net(hyperparameters['vae'])
        self.map = _net(hyperparameters['map'])
        # optimizers / schedulers (lsps_trainer.py:26-34)
        self.dis_opt = FlatAdam(self.dis.parameters(), lr=lr, betas=(0.5, 0.999), weight_decay=0.0001)
        self.gen_opt = FlatAdam(list(self.gen.parameters()) + list(self.map.parameters()), lr=lr, betas=(0.5, 0.999),
                                weight_decay=0.0001)
        self.vae_opt = FlatAdam(self.vae.parameters(), lr=lr * 10., betas=(0.5, 0.999), weight_decay=0.001)
        MS = torch.optim.lr_scheduler.MultiStepLR
        self.dis_sch = MS(self.dis_opt, milestones=[200, 300, 400, 450], gamma=0.5)
        self.gen_sch = MS(self.gen_opt, milestones=[200, 300, 400, 450], gamma=0.5)
        self.vae_sch = MS(self.vae_opt, milestones=[125, 175], gamma=0.1)
        for net in (self.dis, self.gen, self.vae, self.map):                      # :37-40
            net.apply(gaussian_weights_init)
        self.gpu = None
        self._reducers = {}
        self._graphs_on = False
        self._graphs = {}
        self._graph_seen = set()
        self._graph_pool = None
        self._capturing = None
        self._side = None
        self._frozen_decided = None
        self._gen_epoch_last = None
        self._enc_shared_pass = None        # (key, pre): options.share_encoder
        self.math_mode = None               # None: whatever the process-wide mode is (ops.set_math_mode); see set_modes
        self.winograd = None

    # ------------------------------------------------------------------ device / arenas
    def cuda(self, gpu=None):
        """Moves the nets to HIP device `gpu` and builds the flat parameter arenas (:334-347)."""
        if not torch.cuda.is_available():
            raise ops._lib.LspsHipError("LSPSTrainer.cuda(): no HIP device visible; there is no CPU path")
        self.gpu = torch.cuda.current_device() if gpu is None else gpu
        dev = torch.device('cuda', self.gpu)
        torch.cuda.set_device(dev)
        for net in (self.dis, self.gen, self.vae, self.map):
            nn.Module.cuda(net, dev)
        for key, opt, nets in (('dis', self.dis_opt, (self.dis,)), ('gen', self.gen_opt, (self.gen, self.map)),
                               ('vae', self.vae_opt, (self.vae,))):
            arena = opt.attach()
            for n_ in nets:
                n_._arena = arena
            opt.grad_scale = 1.0 / lsps_dist.world()
            seg, starts = 0, []
            for n_ in nets:                 # no bucket may hold gradients of two nets (gen | map share one arena)
                starts.append(seg)
                seg += len(list(n_.parameters()))
            self._reducers[key] = lsps_dist.GradReducer(arena, segments=starts)
        self._drop_graphs()                 # captured launches point into the previous arenas
        self.sync_replicas()
        return self

    def set_modes(self, math_mode=None, winograd=None):
        """This trainer's own math mode ('f32' | 'bf16' | 'f32_split') and conv algorithm ('auto' | 'off' | 'always' | ...),
        installed around each of its update methods and restored afterwards; None = follow the process-wide setting."""
        self.math_mode, self.winograd = math_mode, winograd
        return self

    def _drop_graphs(self):
        self._graphs.clear()
        self._graph_seen.clear()
        self._graph_pool = None

    def sync_replicas(self):
        """Under data parallelism every rank builds its nets from its own RNG (gaussian_weights_init, nn.Linear's
        default init) — the replicas are made identical by broadcasting rank 0's parameter arenas, Adam moments and step
        counts.  Called after `cuda()`, `resume()` and `load_vae()`; no-op in a single process."""
        if self.gpu is None:
            return
        for opt in (self.dis_opt, self.gen_opt, self.vae_opt):
            opt.sync_from_rank0()

    def _declare_frozen(self, method):
        """post_update never steps the generator (the estimate modes train the discriminator / regressor only,
        lsps_trainer.py:220-262): its packed weight panels are kept from step to step (ops.weight_cache_frozen) — once the
        generator's weights have been seen UNCHANGED by two consecutive post_update calls (FlatArena.epoch), i.e. in an
        estimate loop; a workflow that alternates gen_update and post_update keeps packing per step.  The other update
        methods declare nothing frozen.  Returns the epoch in force (or None).  Once per call: `_graphed` decides before it
        looks a graph up, the eager body then finds the decision made."""
        if self._frozen_decided is not None:
            ep, self._frozen_decided = self._frozen_decided[0], None
            return ep
        arena = self.gen_opt.arena
        ep = None
        if method == 'post_update' and arena is not None:
            now = arena.epoch()
            stable, self._gen_epoch_last = (now == self._gen_epoch_last), now
            if stable and ops.weight_cache_frozen(arena, now):
                ep = now
        if ep is None:
            ops.weight_cache_frozen(None)
        return ep

    def _side_stream(self, device):
        """Second HIP stream for the independent branch of the estimate modes (options.overlap = False: none).  Also under data
        parallelism: the gradient hooks run in the AccumulateGrad nodes, which the engine executes on the launch stream after
        it has joined the side stream's producers, so a bucket's all-reduce is ordered behind both branches."""
        if not ops.options.get().overlap:
            return None
        if self._side is None:
            # side_prio = -1: a high-priority stream (its quarter-chip launches are the step's critical chain)
            self._side = torch.cuda.Stream(device=device, priority=ops.options.get().side_prio)
            # the discriminator's AccumulateGrad nodes live on the main stream while part of their gradients now arrive from
            # the side stream: intended (the engine synchronises them), so the advisory warning is switched off
            fn = getattr(torch.autograd.graph, 'set_warn_on_accumulate_grad_stream_mismatch', None)
            if fn is not None:
                fn(False)
        return self._side

    def use_graphs(self, on=True):
        """hipGraph replay of dis_update / gen_update / post_update (under torch.distributed only on the RCCL backend, whose
        collectives can be captured; gloo steps stay eager).  A call signature = method + tensor shapes + every non-tensor argument (mode, the
        hyperparameter dict, feat_mat ...) + train/eval state; its first call runs eagerly (warm-up: workspaces, kernel
        attributes), the second is captured, later ones replay.  Results are those of the eager path (same kernels, same
        order); the tensors a graphed method returns are static buffers overwritten by the next replay.  Random draws
        (GaussianNoiseLayer, the VAE code) come from torch's graph-safe generator offsets."""
        self._graphs_on = bool(on)
        if not on:
            self._drop_graphs()             # a later use_graphs(True) starts a fresh memory pool
        return self

    def _graphed(self, name, eager, args, kwargs):
        tensors = []
        # everything that decides WHICH launches a step makes or WHERE they read / write: argument shapes and values, every
        # sub-net's train / eval flag (poseVAE.encode draws noise, GaussianNoiseLayer is off in eval), the algorithm /
        # math switches of the library, the process options as a whole (lsps_amd/options.py: a frozen, hashable object — a
        # switch added there is part of the signature by construction) and the arenas' addresses
        sig = (name, _flatten_tensors((args, kwargs), tensors), self.gen.training, self.dis.training, self.vae.training,
               self.map.training, ops.get_winograd(), ops.get_math_mode(), ops.options.get())
        oob_tag = sig                            # the rank-independent part: key of the ranks' capture agreement (no addresses)
        sig = sig + (tuple(int(o.arena.flat_p.data_ptr()) for o in (self.dis_opt, self.gen_opt, self.vae_opt)
                           if o.arena is not None),)
        if name == 'post_update':
            # a post_update captured while the generator's panels are frozen holds no pack launches for them (they were cache
            # hits): it is only valid while that cache is — same generator epoch — and such graphs of older epochs can never be
            # replayed again.  (ep None: nothing frozen, the graph packs for itself and is valid for any weights.)
            # The frozen table is ONE per process (csrc/igemm.hip: g_fz): another trainer's post_update re-targets it, after which
            # a later re-pack of THIS trainer's panels may lay them out differently in its buffer.  `frozen_resets` counts those
            # re-targetings (ops.weight_cache_frozen); a frozen graph is only valid for the count it was captured under
            # (two graphed trainers alternating in one process).
            ep = self._declare_frozen(name)
            self._frozen_decided = (ep,)
            tag = None if ep is None else (ep, ops.frozen_resets())
            sig = sig + (('gen_epoch', tag),)
            for old in [k for k in self._graphs if k[0] == 'post_update' and k[-1][1] not in (None, tag)]:
                del self._graphs[old]
            if not self._graphs:
                self._graph_pool = None      # torch releases a private pool with its last graph: never capture into a dead handle
        g = self._graphs.get(sig)
        if g is not None:
            self._frozen_decided = None
            try:
                return g.replay(self, args, kwargs)
            finally:
                ops.weight_cache_frozen(None)   # the promise ends with the method, also on the replay path
        if sig not in self._graph_seen:                  # warm-up call: eager
            self._graph_seen.add(sig)
            return eager(self, *args, **kwargs)
        torch.cuda.synchronize()
        if lsps_dist.active():
            # RCCL's watchdog must have RETIRED every eager collective before its stream joins a capture (dist.drain_watchdog:
            # confirmed through the flight recorder, not a sleep).  Not confirmed (recorder off) on ANY rank => this call
            # stays eager on EVERY rank and the capture is tried again at the next one.  The decision must be the same
            # everywhere (a rank that stayed eager alone would issue, one call later, an agreement collective the replaying
            # ranks do not: ADVICE r4), and the agreement must not itself be RCCL work that needs draining: it goes through
            # the rendezvous store (dist.agree_all_oob).
            if not lsps_dist.agree_all_oob(lsps_dist.drain_watchdog(), tag=oob_tag):
                return eager(self, *args, **kwargs)
        if self._graph_pool is None:
            self._graph_pool = torch.cuda.graph_pool_handle()
        fence = lsps_dist.capture_begin()
        try:
            g = self._graphs[sig] = _GraphedUpdate(self, eager, args, kwargs, self._graph_pool)
        finally:
            lsps_dist.capture_end(fence)
        # the capture only recorded the launches: run them once for this call's result
        try:
            return g.replay(self, args, kwargs)
        finally:
            ops.weight_cache_frozen(None)

    def _step(self, key, opt, loss, names, tensors, sig, begun=False):
        """backward + gradient exchange + optimizer step + publication of the step's scalars (stored as numpy values like
        the reference, ONE device->host copy).  `sig` identifies the graph of this step for the reducer: it learns which
        parameters get gradients under that signature and launches each bucket during backward (lsps_amd/dist.py).  Under
        data parallelism the scalars are summed over ranks by one tiny all-reduce launched BEFORE backward (they are
        forward results), so publishing costs no second host synchronisation.  While a hipGraph is being captured the
        optimizer step and the host copy are left to the replay (`_finish_step`).
        `begun`: the caller opened the step itself (`_begin_backward`) and already ran a first partial backward; `loss`
        is the remaining term."""
        red = self._reducers[key]
        scal = torch.stack([t.detach().reshape(()).float() for t in tensors])
        if begun:
            red.add_scalars(scal if lsps_dist.active() else None)
        else:
            red.begin(sig, scalars=scal if lsps_dist.active() else None)
        try:
            loss.backward()
            red.finish()
        finally:
            ops.weight_cache_end()          # the optimizer is about to change the weights
        if self._capturing is not None:
            self._capturing.pending = (key, opt, list(names), scal)
            return
        self._finish_step(opt, names, scal, red.reduced_scalars())

    def _encoder_key(self, images_a, images_b):
        """What the encoder pass depends on, or None when sharing it is off / impossible: hipGraph replay (a tape cannot cross two
        captures), residual-block dropout (the reference draws a fresh mask per pass), generators without `encode_pre`."""
        if not ops.options.get().share_encoder or self._graphs_on or not hasattr(self.gen, 'encode_pre'):
            return None
        if any(getattr(m, 'dropout', 0) for m in self.gen.modules() if hasattr(m, 'dropout') and isinstance(getattr(m, 'dropout'), float)):
            return None
        arena = self.gen_opt.arena
        return (images_a.data_ptr(), images_a._version, tuple(images_a.shape), images_b.data_ptr(), images_b._version,
                tuple(images_b.shape), arena.epoch() if arena is not None else None, self.gen.training,
                ops.get_math_mode(), ops.get_winograd(), ops.options.get())

    def _begin_backward(self, key, sig):
        """Opens the gradient exchange of a step whose loss terms are differentiated one by one (post_update)."""
        self._reducers[key].begin(sig)

    def _finish_step(self, opt, names, scal, mean):
        opt.step()
        src = scal if mean is None else mean
        if not src.is_cuda or not self.lazy_scalars or any(k not in _SCALAR_NAMES for k in names) or len(names) > 32:
            vals = src.cpu().numpy()                                 # (CPU stand-ins of the tests; opt-out: trainer.lazy_scalars = False)
            for k, v in zip(names, vals):
                setattr(self, k, np.asarray(v, dtype=np.float32))
            return
        # asynchronous publication: next slot of the ring (values still living in it are materialised first: their copy finished
        # a ring's length of steps ago), one non-blocking copy, an event; the properties wait for that event on first read
        ring = self.__dict__.setdefault('_scalar_ring', [])
        if len(ring) < 8:
            ring.append(_ScalarSlot())
            slot = ring[-1]
        else:
            slot = ring[self._scalar_next % 8]
            for k in slot.names:
                v = self.__dict__.get(k)
                if isinstance(v, tuple) and v[0] is slot:
                    getattr(self, k)
        self.__dict__['_scalar_next'] = self.__dict__.get('_scalar_next', 0) + 1
        slot.host[:len(names)].copy_(src.detach().float().reshape(-1), non_blocking=True)
        slot.event.record()
        slot.names = list(names)
        for i, k in enumerate(names):
            self.__dict__[k] = (slot, i)

    def __getstate__(self):
        # copy.deepcopy / pickle: scalars still in flight become plain numpy values; the pinned ring (events cannot be copied) stays behind
        for k in _SCALAR_NAMES:
            if isinstance(self.__dict__.get(k), tuple):
                getattr(self, k)
        d = self.__dict__.copy()
        d.pop('_scalar_ring', None)
        d.pop('_scalar_next', None)
        return d

    def __dir__(self):
        # the reference's write_loss reflects over dir(trainer) (common.py:73-80): scalars no update method has published yet do not exist
        return sorted(set(k for k in super(LSPSTrainer, self).__dir__() if k not in _SCALAR_NAMES or k in self.__dict__))

    # ------------------------------------------------------------------ loss helpers (:42-60)
    def _compute_ll_loss(self, a, b):
        return ops.l1_loss(a, b)

    def _compute_l2_loss(self, a, b):
        return ops.l2_loss(a, b)

    def _compute_kl(self, mu, sd=None):
        return ops.kl_loss(mu, sd)

    # ------------------------------------------------------------------ :62-74
    def vae_update(self, y, hyperparameters, noise=None):
        self.vae.zero_grad()
        dec, z, mu, sd = self.vae(y, noise=noise)
        enc_loss = self._compute_kl(mu, sd)
        ll_loss = self._compute_ll_loss(dec, y)
        total_loss = hyperparameters['kl_loss_vae'] * enc_loss + hyperparameters['ll_loss_vae'] * ll_loss
        self._step('vae', self.vae_opt, total_loss, ['vae_total_loss'], [total_loss], ('vae_update',))
        return dec

    def _pose2depth(self, labels_a, labels_b, nz):
        """labels -> (noisy) pose code -> Mapping -> gen.decode; first half of decode_A, second half of
        decode_B (lsps_trainer.py:87-93,148-154).  The pose code is a constant: vae_opt never steps here."""
        with torch.no_grad():
            enc_pose, _, _ = self.vae.encode(torch.cat((labels_a, labels_b), 0), noise=nz)
        z = self.map(enc_pose)
        dec_A, dec_B = self.gen.decode(z)
        half = dec_A.size(0) // 2
        return z, dec_A[:half], dec_B[half:]

    # ------------------------------------------------------------------ :76-141
    @_weights_constant
    def gen_update(self, images_a, labels_a, images_b, labels_b, hyperparameters, noise=(None, None, None)):
        hp = hyperparameters
        self.gen.zero_grad()                                   # one arena: also zeroes the Mapping grads (:85)
        pre = None
        if self._enc_shared_pass is not None:
            key, cached = self._enc_shared_pass
            self._enc_shared_pass = None
            if key == self._encoder_key(images_a, images_b):
                pre = cached
        x_aa, x_ba, x_ab, x_bb, shared = self.gen(images_a, images_b, noise=noise[0], pre=pre)
        x_bab, shared_bab = self.gen.forward_a2b(x_ba, noise=noise[1])
        x_aba, shared_aba = self.gen.forward_b2a(x_ab, noise=noise[2])
        decode_A, decode_B, data_a, data_b = x_ba, x_ab, x_ba, x_ab
        map_terms = None
        if hp['train_map']:                                                       # :84-100
            z_p2d, decode_A, decode_B = self._pose2depth(labels_a, labels_b, noise[3] if len(noise) > 3 else None)
            data_a, data_b = torch.cat((x_ba, decode_A), 0), torch.cat((x_ab, decode_B), 0)
            map_terms = (self._compute_l2_loss(shared, z_p2d),
                         self._compute_ll_loss(decode_A, images_a) + self._compute_ll_loss(decode_B, images_b))
        with self.dis.frozen():
            outs_a, outs_b, _, _ = self.dis(data_a, data_b)
        ad_loss_a, _ = ops.bce_sigmoid(outs_a, 1.0)
        ad_loss_b, _ = ops.bce_sigmoid(outs_b, 1.0)
        enc_loss = self._compute_kl(shared)
        enc_bab_loss, enc_aba_loss = self._compute_kl(shared_bab), self._compute_kl(shared_aba)
        ll_loss_a, ll_loss_b = self._compute_ll_loss(x_aa, images_a), self._compute_ll_loss(x_bb, images_b)
        ll_loss_aba, ll_loss_bab = self._compute_ll_loss(x_aba, images_a), self._compute_ll_loss(x_bab, images_b)
        total_loss = hp['gan_w'] * (ad_loss_a + ad_loss_b) + \
            hp['ll_direct_link_w'] * (ll_loss_a + ll_loss_b) + \
            hp['ll_cycle_link_w'] * (ll_loss_aba + ll_loss_bab) + \
            hp['kl_direct_link_w'] * (enc_loss + enc_loss) + \
            hp['kl_cycle_link_w'] * (enc_bab_loss + enc_aba_loss)                 # enc_loss doubled as in :124
        names = ['gen_enc_loss', 'gen_enc_loss2', 'gen_ad_loss', 'gen_ll_loss', 'gen_ll_loss2']
        vals = [enc_loss, enc_aba_loss + enc_bab_loss, ad_loss_a + ad_loss_b, ll_loss_a + ll_loss_b,
                ll_loss_bab + ll_loss_aba]
        if map_terms is not None:
            total_loss = total_loss + hp['ll_map_z_w'] * map_terms[0] + hp['ll_map_w'] * map_terms[1]
            names += ['gen_map_loss', 'gen_map_loss2']
            vals += [map_terms[0], map_terms[1]]
        self._step('gen', self.gen_opt, total_loss, names + ['gen_total_loss'], vals + [total_loss],
                   ('gen_update', bool(hp['train_map']), self.gen.training))
        return (x_aa, x_ba, x_ab, x_bb, x_aba, x_bab, decode_A, decode_B)

    # ------------------------------------------------------------------ :143-218
    @_weights_constant
    def dis_update(self, images_a, labels_a, images_b, labels_b, com_a, com_b, hyperparameters, feat_mat=True,
                   noise=None):
        hp = hyperparameters
        self.dis.zero_grad()
        nz_gen, nz_vae = (noise if isinstance(noise, (tuple, list)) else (noise, None))
        self._enc_shared_pass = None
        key = self._encoder_key(images_a, images_b)
        if key is not None:
            # opt-in (options.share_encoder): the encoder half of gen(images_a, images_b) is the same function of the same images and
            # weights in this call and in the gen_update behind it (the two differ in the noise draw only): it runs ONCE, with its
            # tape, and gen_update continues from it.  Identical results (the same kernels as gen_update's own encoder pass).
            with torch.enable_grad():
                pre = self.gen.encode_pre(images_a, images_b)
            self._enc_shared_pass = (key, pre)
        with torch.no_grad():
            x_aa, x_ba, x_ab, x_bb, _ = self.gen(images_a, images_b, noise=nz_gen, pre=pre.detach() if key is not None else None)
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
        names, vals = [], []
        feature_loss = None
        if feat_mat:                                                              # :171-177
            feat_as = torch.split(feats_a, feats_a.size(0) // ndiv, 0)
            feat_bs = torch.split(feats_b, feats_a.size(0) // ndiv, 0)
            feature_loss = self._compute_ll_loss(feat_bs[1], feat_as[2]) + self._compute_ll_loss(feat_as[1], feat_bs[2])
        ra = torch.split(res_a, res_a.size(0) // ndiv, 0)
        rb = torch.split(res_b, res_b.size(0) // ndiv, 0)
        true_a, cnt_ta = ops.bce_sigmoid(ra[0], 1.0)                              # :189-192
        true_b, cnt_tb = ops.bce_sigmoid(rb[0], 1.0)
        fake_a, cnt_fa = ops.bce_sigmoid(ra[1], 0.0)
        fake_b, cnt_fb = ops.bce_sigmoid(rb[1], 0.0)
        n_true, n_fake = float(ra[0].numel()), float(ra[1].numel())
        true_acc = 0.5 * (cnt_ta[0] + cnt_tb[0]) / n_true                         # helpers.py:20-32, :194-199
        fake_acc = 0.5 * (cnt_fa[1] + cnt_fb[1]) / n_fake
        ad_loss = (true_a + fake_a) + (true_b + fake_b)
        if hp['train_map']:                                                       # :201-204 fake decoded-from-pose term
            ad_loss = ad_loss + ops.bce_sigmoid(ra[3], 0.0)[0] + ops.bce_sigmoid(rb[3], 0.0)[0]
        loss = hp['gan_w'] * ad_loss
        if feat_mat:
            loss = loss + hp['feature_w'] * feature_loss
        names = ['dis_ad_loss', 'dis_loss', 'dis_true_acc', 'dis_fake_acc']
        vals = [ad_loss, loss, true_acc, fake_acc]
        if feat_mat:
            names.append('dis_feat_loss')
            vals.append(feature_loss)
        self._step('dis', self.dis_opt, loss, names, vals, ('dis_update', bool(feat_mat), bool(hp['train_map'])))
        return

    # ------------------------------------------------------------------ :220-262
    @_weights_constant
    def post_update(self, images_a, labels_a, images_b, labels_b, com_a, com_b, mode, hyperparameters, noise=None):
        hp = hyperparameters
        noise = noise or {}
        self.dis.zero_grad()
        x_aa, x_ba, x_ab, x_bb = images_a, images_a, images_b, images_b
        terms_reg, terms_feat = [], []

        def regression(front, images, labels, nz):
            _, pred, _ = front(images)
            with torch.no_grad():                                                 # target: noisy vae code (:229,:246)
                target, _, _ = self.vae.encode(labels, noise=nz)
            return self._compute_l2_loss(pred, target.reshape(pred.shape))

        if mode == 0:
            terms_reg.append(regression(self.dis.regress_a, images_a, labels_a, noise.get('vae_a')))
        elif mode == 1:
            terms_reg.append(regression(self.dis.regress_b, images_b, labels_b, noise.get('vae_b')))
        else:
            # :238 — only the first 4 samples OF THE GLOBAL BATCH: under data parallelism every rank evaluates the SAME feature
            # term (exact global-batch parity), wherever those samples live (dist.global_first: rank 0 alone when the shard
            # holds >= 4, the first ranks' shards otherwise)
            first_a, first_b = lsps_dist.global_first((images_a, images_b), 4)
            if ops.options.get().est_merge:
                # Round 5: ONE pass of the discriminator over [regression samples | the 16 feature samples] (SharedDis.regress_feats).
                # The generator pass on the first 4 + 4 samples has no consumer but that pass, so nothing is left to overlap it
                # with except the host: it runs on the launch stream.  Per-sample identical to the two separate calls; every
                # discriminator weight gets ONE gradient contribution (data parallel: the reducer's plain learned set).
                with torch.no_grad():
                    x_aa, x_ba, x_ab, x_bb = self.gen(first_a, first_b, noise=noise.get('gen'))[:4]
                post_a, post_b, (f_x_aa, f_x_ba, f_x_ab, f_x_bb) = self.dis.regress_feats(
                    images_a, images_b if mode == 4 else None, x_aa, x_ba, x_ab, x_bb)

                def reg_term(pred, labels, nz):
                    with torch.no_grad():                                         # target: noisy vae code (:229,:246)
                        target, _, _ = self.vae.encode(labels, noise=nz)
                    return self._compute_l2_loss(pred, target.reshape(pred.shape))
                reg_loss = reg_term(post_a, labels_a, noise.get('vae_a'))
                if mode == 4:
                    reg_loss = reg_loss + reg_term(post_b, labels_b, noise.get('vae_b'))
                feat_loss = self._compute_ll_loss(f_x_ab, f_x_aa) + self._compute_ll_loss(f_x_ba, f_x_bb)
                total_loss = hp['reg_w'] * reg_loss + hp['feature_w_reg'] * feat_loss
                self._step('dis', self.dis_opt, total_loss, ['dis_reg_loss', 'dis_total_loss'], [reg_loss, total_loss],
                           ('post_update', int(mode), 'merged'))
                return (x_aa, x_ba, x_ab, x_bb, x_aa, x_bb, x_aa, x_bb)
            # (est_merge off: the round-4 schedule)
            # The feature branch (generator on 8 samples -> dis.feats on 16) and the regression branch (dis on the whole
            # batch) are independent until the loss is summed, and the first one's launches fill a quarter of the chip at
            # best: it runs on a second HIP stream (forward here, its backward follows it there: autograd replays a node on
            # the stream of its forward).  Same kernels, same arithmetic; per-stream workspaces and pack-cache entries.
            side = self._side_stream(images_a.device)
            main = torch.cuda.current_stream(images_a.device)
            # Schedule (profiles/r4*_estimate3_timeline_*.txt).  The two branches share nothing but the discriminator's weights,
            # and the loss is their weighted SUM, so d loss = reg_w d reg + feature_w_reg d feat is taken one term at a time:
            # the regression term is differentiated right behind its own forward instead of waiting for the generator pass
            # of the other branch (with ONE backward over the summed loss nothing of either backward can start before both
            # forwards have ended).  The two partial gradients meet in the parameters' AccumulateGrad nodes: each
            # discriminator weight gets one contribution per term, and a + b = b + a in floating point, so the result is
            # bit-identical to the single backward (tests/test_parity_gpu.py: overlapped == serial, bitwise).
            split = side is not None and ops.options.get().est_split_backward
            sig = ('post_update', int(mode))

            def regression_branch():
                terms_reg.append(regression(self.dis.regress_a, images_a, labels_a, noise.get('vae_a')))
                if mode == 4:
                    terms_reg.append(regression(self.dis.regress_b, images_b, labels_b, noise.get('vae_b')))
            if side is not None:
                side.wait_stream(main)
            def regression_backward():
                self._begin_backward('dis', sig)
                (hp['reg_w'] * sum(terms_reg[1:], terms_reg[0])).backward()

            def feature_forward():
                with torch.cuda.stream(side if side is not None else main):
                    with torch.no_grad():
                        outs = self.gen(first_a, first_b, noise=noise.get('gen'))[:4]
                    f_x_aa, f_x_ba, f_x_ab, f_x_bb = self.dis.feats(*outs)
                    terms_feat.append(self._compute_ll_loss(f_x_ab, f_x_aa))
                    terms_feat.append(self._compute_ll_loss(f_x_ba, f_x_bb))
                return outs
            # Launch / capture order of the branches (measured, profiles/r4f_estimate3_orders.txt; whichever branch is launched
            # first runs almost alone for its first ~2.5 ms — its kernels fill the chip — so the order decides what the tail is):
            #   'feat_first' [generator pass, feature forward | regression forward + backward], feature backward   6.88 ms (default)
            #   'reg_first'  [regression forward + backward | generator pass, feature forward], feature backward   7.18 ms
            #   'chain'      regression forward, THEN [generator pass, feature forward | regression backward], ...  7.17 ms
            order = ops.options.get().est_order if split else 'serial'
            if order == 'chain':
                regression_branch()
                side.wait_stream(main)
                x_aa, x_ba, x_ab, x_bb = feature_forward()
                regression_backward()
            elif order == 'reg_first':
                regression_branch()
                regression_backward()
                x_aa, x_ba, x_ab, x_bb = feature_forward()
            else:
                x_aa, x_ba, x_ab, x_bb = feature_forward()
                if split:
                    regression_branch()
                    regression_backward()
            if split:
                reg_loss = sum(terms_reg[1:], terms_reg[0]).detach()
            if not split:
                regression_branch()
            if side is not None:
                main.wait_stream(side)
            if split:
                feat_loss = hp['feature_w_reg'] * (terms_feat[0] + terms_feat[1])
                total_loss = hp['reg_w'] * reg_loss + feat_loss
                self._step('dis', self.dis_opt, feat_loss, ['dis_reg_loss', 'dis_total_loss'], [reg_loss, total_loss], sig,
                           begun=True)
                return (x_aa, x_ba, x_ab, x_bb, x_aa, x_bb, x_aa, x_bb)
        reg_loss = sum(terms_reg[1:], terms_reg[0])
        total_loss = hp['reg_w'] * reg_loss
        if terms_feat:
            total_loss = total_loss + hp['feature_w_reg'] * (terms_feat[0] + terms_feat[1])
        self._step('dis', self.dis_opt, total_loss, ['dis_reg_loss', 'dis_total_loss'], [reg_loss, total_loss],
                   ('post_update', int(mode)))
        return (x_aa, x_ba, x_ab, x_bb, x_aa, x_bb, x_aa, x_bb)

    # ------------------------------------------------------------------ :264-276, 349-350
    def normalize_image(self, x):
        return x[:, 0:3, :, :]

    def assemble_outputs(self, images_a, images_b, network_outputs):
        o = [self.normalize_image(t) for t in network_outputs]
        a, b = self.normalize_image(images_a), self.normalize_image(images_b)
        x_aa, x_ba, x_ab, x_bb, x_aba, x_bab, dec_a, dec_b = o
        return torch.cat((a[0:1], x_aa[0:1], x_ab[0:1], x_aba[0:1], dec_a[0:1], dec_b[0:1],
                          b[0:1], x_bb[0:1], x_ba[0:1], x_bab[0:1]), 3)

    # ------------------------------------------------------------------ checkpoints (:278-332)
    @staticmethod
    def _load(path):
        return torch.load(path, map_location='cpu')

    # what a rank may fail with when ITS copy of the snapshot is missing or unreadable (no shared filesystem, a file another
    # rank is still writing): torch.load on a truncated zip raises RuntimeError('PytorchStreamReader failed ...'), a cut
    # pickle UnpicklingError / EOFError.  Anything else is a configuration error (state-dict key or shape mismatch).
    _ABSENT = (IndexError, FileNotFoundError, OSError, EOFError, _pickle.UnpicklingError)

    @classmethod
    def _snapshot_absent(cls, ex):
        if isinstance(ex, cls._ABSENT):
            return True
        return isinstance(ex, RuntimeError) and ('PytorchStreamReader' in str(ex) or 'zip archive' in str(ex)
                                                 or 'unexpected EOF' in str(ex))

    def resume(self, snapshot_prefix, idx=-1, load_opt=False, est=False):
        """Under data parallelism the outcome is rank 0's: a rank that does not see the snapshot (no shared filesystem, a
        half-written file) still enters the same collectives as the others and ends up with rank 0's weights and count.
        EVERY rank records whatever its local load raised and then enters the same two agreements (a rank that
        raises alone leaves the others blocked in a broadcast until the launcher kills them):
        1. rank 0's outcome — if rank 0 failed, all ranks raise together;
        2. whether any OTHER rank failed with a configuration error (not a missing / unreadable file) — then all ranks
           raise together too; a rank whose copy was merely absent adopts rank 0's weights in `sync_replicas()`."""
        err = None
        try:
            iterations = self._resume_local(snapshot_prefix, idx, load_opt, est)
        except Exception as ex:                     # noqa: BLE001  (kept and re-raised below, on every rank together)
            if lsps_dist.world() == 1:
                raise                               # single process: the reference's behaviour (helpers.get_model_list)
            err, iterations = ex, 0
        if lsps_dist.world() > 1:
            ok0, iterations = lsps_dist.agree_from_rank0((err is None, iterations))
            if not ok0:
                raise err if (err is not None and lsps_dist.rank() == 0) else RuntimeError(
                    "resume(): rank 0 could not load the snapshot '%s'" % snapshot_prefix)
            config_error = err is not None and not self._snapshot_absent(err)
            if not lsps_dist.agree_all(not config_error):
                raise err if config_error else RuntimeError(
                    "resume(): another rank could not apply the snapshot '%s' (state-dict mismatch)" % snapshot_prefix)
        self.sync_replicas()
        return iterations

    def _resume_local(self, snapshot_prefix, idx, load_opt, est):
        dirname = os.path.dirname(snapshot_prefix)
        last_model_name = get_model_list(dirname, "est_gen" if est else "gen", idx)
        if last_model_name is None:
            return 0
        self.gen.load_state_dict(self._load(last_model_name), strict=False)
        iterations = int(last_model_name[-12:-4])
        last_model_name = get_model_list(dirname, "est_dis" if est else "dis", idx)
        self.dis.load_state_dict(self._load(last_model_name), strict=False)
        if load_opt:
            try:
                self.gen_opt.load_state_dict(self._load(get_model_list(dirname, "optg", idx)))
                self.dis_opt.load_state_dict(self._load(get_model_list(dirname, "optd", idx)))
                print('-----optimizer parameters loaded!')
            except Exception:
                print('-----Failed to load optimizer parameters!')
        try:
            self.map.load_state_dict(self._load(get_model_list(dirname, "map", idx)), strict=False)
        except Exception:
            print('-----Failed to load map parameters!')
        print('Resume from iteration %d' % iterations)
        return iterations

    @staticmethod
    def _dense_state(net):
        return dict((k, v.detach().clone()) for k, v in net.state_dict().items())   # not views of the arena

    def save(self, snapshot_prefix, iterations):
        """Replicas are identical, so rank 0 alone writes (N ranks writing one path would race); the barrier keeps a
        following resume() on another rank from reading a half-written file."""
        if lsps_dist.rank() == 0:
            torch.save(self._dense_state(self.gen), '%s_gen_%08d.pkl' % (snapshot_prefix, iterations + 1))
            torch.save(self._dense_state(self.dis), '%s_dis_%08d.pkl' % (snapshot_prefix, iterations + 1))
        lsps_dist.barrier()

    def save_vae(self, snapshot_prefix, iterations, frac):
        if lsps_dist.rank() == 0:
            torch.save(self._dense_state(self.vae), '%s_vae_%.2f_%08d.pkl' % (snapshot_prefix, frac, iterations + 1))
        lsps_dist.barrier()

    def load_vae(self, snapshot_prefix, frac):
        dirname = os.path.dirname(snapshot_prefix)
        last_model_name = get_model_list(dirname, 'vae_%.2f' % frac)
        if last_model_name is not None:
            self.vae.load_state_dict(self._load(last_model_name))
            print('Loading pretrained VAE parameters from %s' % last_model_name)
        self.sync_replicas()                # every rank, whatever it found locally: rank 0's copy wins
        return 0


# ---------------------------------------------------------------------------------------------
# harness hooks used by tests/golden/cases.py (NativeAdapter)
# ---------------------------------------------------------------------------------------------
def make_trainer(hp, device='cuda'):
    tr = LSPSTrainer(hp)
    dev = torch.device(device)
    tr.cuda(dev.index if dev.index is not None else torch.cuda.current_device())
    return tr


def set_training(gen, flag):
    gen.train(bool(flag))


def named_grads(net, to_numpy):
    arena = net._arena
    touched = {}
    if arena is not None:
        touched = dict((id(p), t) for p, t in zip(arena.params, arena.touched))
    out = {}
    for k, p in net.named_parameters():
        out[k] = to_numpy(p.grad) if (p.grad is not None and touched.get(id(p), True)) else None
    return out


for _k in _SCALAR_NAMES:
    setattr(LSPSTrainer, _k, _scalar_property(_k))
del _k
