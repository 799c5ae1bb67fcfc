"""Flat parameter / gradient arena and the fused Adam step over it.

MI355X-first layout: all parameters of one optimizer live back-to-back in ONE HBM buffer, their
gradients in a second one and the two Adam moments in two more.  Consequences:
  * the optimizer step is a single kernel launch (lsps_adam_step) over the arena instead of ~100
    per-tensor launches (reference: torch.optim.Adam, src/trainers/lsps_trainer.py:26-29);
  * data-parallel gradient exchange is an all-reduce of a few large contiguous slices of the
    gradient buffer (lsps_amd/dist.py) — no flatten/unflatten copies;
  * `zero_grad` is one memset.
`nn.Parameter.data` and `.grad` are views into the arenas, so autograd accumulates in place and
state dicts / checkpoints see ordinary tensors with the reference's keys.

Semantics kept from torch.optim.Adam (torch 2.x, as captured in the golden vectors): coupled L2
weight decay, bias correction with a PER-PARAMETER step count, and parameters that received no
gradient in this step are skipped entirely (torch: `p.grad is None`).
"""
import math

import numpy as np
import torch

from . import _lib


class FlatArena(object):
    """Re-homes `params` (already on the target device) into flat p/g buffers."""
    _uids = 0

    def __init__(self, params):
        self.params = [p for p in params]
        assert self.params, "empty parameter list"
        dev = self.params[0].device
        self.device = dev
        self.offsets, off = [], 0
        for p in self.params:
            assert p.device == dev and p.dtype == torch.float32
            self.offsets.append(off)
            off += (p.numel() + 3) // 4 * 4            # keep every tensor 16-byte aligned
        self.total = off
        self.flat_p = torch.zeros(off, dtype=torch.float32, device=dev)
        self.flat_g = torch.zeros(off, dtype=torch.float32, device=dev)
        self.touched = [False] * len(self.params)
        self.generation = 0                             # bumped whenever flat_p is written behind torch's back (Adam kernel,
        self._hooks = []                                # replica broadcast): see epoch()
        FlatArena._uids += 1                            # process-unique: a later arena may be handed the SAME device addresses
        self.uid = FlatArena._uids                      # by the caching allocator, with equal generation and version counts
        for i, (p, o) in enumerate(zip(self.params, self.offsets)):
            view = self.flat_p[o:o + p.numel()].view(p.shape)
            view.copy_(p.data)
            p.data = view
            p.grad = self.flat_g[o:o + p.numel()].view(p.shape)
            self._hooks.append(p.register_post_accumulate_grad_hook(self._make_hook(i)))
            p._lsps_arena = self                        # writers that go through `p.data` announce themselves: mark_dirty()
        self.on_grad_ready = None                       # set by dist.GradReducer
        # (bias index, weight index) of convs whose bias is mathematically dead (trainers/common_net.py:_mark_dead_bias)
        index = dict((id(p), i) for i, p in enumerate(self.params))
        self.dead_bias = [(i, index[id(p._lsps_dead_of)]) for i, p in enumerate(self.params)
                          if getattr(p, '_lsps_dead_of', None) is not None and id(p._lsps_dead_of) in index]

    def _make_hook(self, i):
        def hook(param):
            self.touched[i] = True
            if self.on_grad_ready is not None:
                self.on_grad_ready(i)
        return hook

    def epoch(self):
        """Change counter of the parameter VALUES: the arena's own writers bump `generation`; writers that go through the
        Parameter objects (load_state_dict, an in-place op under no_grad) bump torch's per-tensor version counters.  Used to
        keep packed weight panels across steps while a net is frozen (ops.weight_cache_frozen).  (A write through `p.data`
        is invisible to both: call `arena.generation += 1` after one.)  The arena's process-unique id is part of the value:
        two arenas never share an epoch, even when the allocator gives the second one the first one's addresses."""
        return ((self.uid & 0xffff) << 48) | ((self.generation & 0xffffff) << 24) | (sum(p._version for p in self.params) & 0xffffff)

    def mark_dirty(self):
        """To be called after a write through `p.data` (which neither `generation` nor torch's version counters see):
        `gaussian_weights_init` does it itself; INTEGRATION.md states the contract for user code."""
        self.generation += 1

    def zero_grad(self):
        self.flat_g.zero_()
        for i, (p, o) in enumerate(zip(self.params, self.offsets)):
            self.touched[i] = False
            if p.grad is None or p.grad.data_ptr() != self.flat_g.data_ptr() + 4 * o:
                p.grad = self.flat_g[o:o + p.numel()].view(p.shape)

    def grad_slice(self, i0, i1):
        """Contiguous gradient slice covering parameters i0..i1-1."""
        end = self.offsets[i1] if i1 < len(self.params) else self.total
        return self.flat_g[self.offsets[i0]:end]


class FlatAdam(torch.optim.Optimizer):
    """torch.optim.Adam drop-in (same constructor keys / state_dict layout) that steps a FlatArena
    with ONE lsps_adam_step launch.  The arena is created by `attach()` once the parameters are on
    the HIP device; stepping without it raises (no CPU fallback)."""

    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0):
        defaults = dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay)
        super(FlatAdam, self).__init__(params, defaults)
        assert len(self.param_groups) == 1, "one param group per FlatAdam"
        self.arena = None
        self._copied = None
        self.grad_scale = 1.0                           # 1/world_size under data parallelism

    def attach(self):
        params = self.param_groups[0]['params']
        if not params[0].is_cuda:
            raise _lib.LspsHipError("FlatAdam.attach(): parameters must be on the HIP device first")
        # moments loaded BEFORE the arena exists (driver order: resume(load_opt=True), then cuda()): `self.state[p]` is
        # rebound to arena views below, so take the loaded tensors out first (a shallow copy of each per-param dict)
        old = {p: dict(self.state.get(p) or {}) for p in params}
        self.arena = FlatArena(params)
        a = self.arena
        self.flat_m = torch.zeros_like(a.flat_p)
        self.flat_v = torch.zeros_like(a.flat_p)
        n = len(params)
        self._seg_off = torch.tensor(a.offsets, dtype=torch.int64, device=a.device)
        self._host = torch.empty((3, n), dtype=torch.float32).pin_memory() if torch.cuda.is_available() else \
            torch.empty((3, n), dtype=torch.float32)
        self._dev = torch.empty((3, n), dtype=torch.float32, device=a.device)
        self._seg_len_dev = torch.empty(n, dtype=torch.int32, device=a.device)
        self._seg_len_host = torch.empty(n, dtype=torch.int32)
        if torch.cuda.is_available():
            self._seg_len_host = self._seg_len_host.pin_memory()
        for p, o in zip(params, a.offsets):
            st = self.state[p]
            prev = old.get(p) or {}
            st['step'] = int(prev.get('step', 0)) if not torch.is_tensor(prev.get('step', 0)) else int(prev['step'].item())
            st['exp_avg'] = self.flat_m[o:o + p.numel()].view(p.shape)
            st['exp_avg_sq'] = self.flat_v[o:o + p.numel()].view(p.shape)
            if 'exp_avg' in prev:
                st['exp_avg'].copy_(prev['exp_avg'])
                st['exp_avg_sq'].copy_(prev['exp_avg_sq'])
        return self.arena

    def sync_from_rank0(self):
        """Data-parallel replicas must hold the SAME weights, Adam moments and step counts: rank 0's arena is
        broadcast (three flat buffers + one int vector).  No-op outside a process group."""
        from . import dist as ldist
        if not ldist.active() or self.arena is None:
            return
        params = self.param_groups[0]['params']
        steps = torch.tensor([int(self.state[p]['step']) for p in params], dtype=torch.int64, device=self.arena.device)
        ldist.broadcast_from_rank0([self.arena.flat_p, self.flat_m, self.flat_v, steps])
        self.arena.generation += 1
        for p, t in zip(params, steps.tolist()):
            self.state[p]['step'] = int(t)

    @torch.no_grad()
    def step(self, closure=None):
        if self.arena is None:
            raise _lib.LspsHipError("FlatAdam.step(): no arena — call trainer.cuda(gpu) first (no CPU fallback)")
        g = self.param_groups[0]
        b1, b2 = g['betas']
        a = self.arena
        params = g['params']
        for bi, wi in a.dead_bias:                      # zero gradient (the arena was zeroed) + weight decay, like the reference
            if a.touched[wi]:
                a.touched[bi] = True
        if self._copied is not None:
            self._copied.synchronize()                  # previous step's H2D of the pinned tables
        max_len = 0
        lens = self._seg_len_host
        host = self._host
        for i, p in enumerate(params):
            if a.touched[i]:
                st = self.state[p]
                st['step'] = int(st['step']) + 1
                t = st['step']
                lens[i] = p.numel()
                host[0, i] = 1.0 - b1 ** t
                host[1, i] = math.sqrt(1.0 - b2 ** t)
                max_len = max(max_len, p.numel())
            else:
                lens[i] = 0
                host[0, i] = 1.0
                host[1, i] = 1.0
        if max_len == 0:
            return None
        self._dev.copy_(host, non_blocking=True)
        self._seg_len_dev.copy_(lens, non_blocking=True)
        self._copied = torch.cuda.Event()
        self._copied.record()
        L = _lib.lib()
        _lib.check(L.lsps_adam_step(a.flat_p.data_ptr(), a.flat_g.data_ptr(), self.flat_m.data_ptr(),
                                    self.flat_v.data_ptr(), self._seg_off.data_ptr(), self._seg_len_dev.data_ptr(),
                                    self._dev[0].data_ptr(), self._dev[1].data_ptr(), len(params), int(max_len),
                                    float(g['lr']), float(b1), float(b2), float(g['eps']), float(g['weight_decay']),
                                    float(self.grad_scale), _lib.stream()), 'adam_step')
        a.generation += 1
        return None

    def zero_grad(self, set_to_none=True):
        if self.arena is not None:
            self.arena.zero_grad()
        else:
            super(FlatAdam, self).zero_grad(set_to_none=set_to_none)

    def load_state_dict(self, state_dict):
        """Accepts a torch.optim.Adam / FlatAdam state dict (reference: lsps_trainer.py:288-295, load_opt=True).
        With an arena attached, the loaded moments are copied INTO the flat buffers (the kernel only reads those)."""
        super(FlatAdam, self).load_state_dict(state_dict)
        if self.arena is None:
            return                                   # attach() picks the loaded state up later
        a = self.arena
        for p, o in zip(self.param_groups[0]['params'], a.offsets):
            st = self.state.get(p)
            if not st:
                continue
            n = p.numel()
            for key, flat in (('exp_avg', self.flat_m), ('exp_avg_sq', self.flat_v)):
                view = flat[o:o + n].view(p.shape)
                if key in st and st[key].data_ptr() != view.data_ptr():
                    view.copy_(st[key])
                st[key] = view
            st['step'] = int(st['step'].item()) if torch.is_tensor(st.get('step')) else int(st.get('step', 0))
