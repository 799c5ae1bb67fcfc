"""Data-parallel gradient exchange for the depth path (NEW: the reference is single-GPU).

One process per GPU (`torch.distributed`, backend "nccl" = RCCL over xGMI).  The path shards
naturally along the batch: InstanceNorm has no batch statistics and every loss is a batch mean,
so rank-local mean-loss gradients averaged over ranks ARE the global-batch gradients
(SURVEY.md §8(e)).  The only exchange step is one sum-all-reduce of the gradient arena per
optimizer step; 1/world is folded into the Adam kernel (`FlatAdam.grad_scale`).

xGMI is point-to-point (7 links x ~153 GB/s per GPU), so the arena is cut into a FEW LARGE
contiguous buckets (default 16 MiB) — per-link-bound ring steps want big messages — ordered so
that the bucket whose gradients are produced FIRST by backward (the last layers) is launched
first, on RCCL's own stream, while backward keeps producing the rest (overlap).  Buckets are
slices of the flat gradient buffer: no flatten / unflatten copies.

The reducer is backend-agnostic (works on gloo/CPU tensors), which is how the N>1 path is
covered by CPU tests (tests/test_dist_cpu.py, world_size 2).
"""
import torch
import torch.distributed as dist


# 16 MiB: few, large messages (an 8-rank ring moves 2 MiB pieces per step: per-link-bound, not latency-bound), yet small
# enough that what is still un-sent when backward ends — the last bucket, i.e. the FIRST layers — stays a few MB: on the
# discriminator that is 6.5 MB (a 32 MiB cut leaves model_S.2's 19 MB weight in it: 25 MB exposed per estimate step)
from . import options
from .options import DEFAULT_BUCKET_BYTES  # noqa: F401


def world():
    return dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1


def active():
    """True when gradients must be exchanged.  LSPS_FORCE_DP=1 also runs the exchange in a 1-rank process
    group (used to smoke-test the RCCL call pattern on a single-GPU box)."""
    if not (dist.is_available() and dist.is_initialized()):
        return False
    return dist.get_world_size() > 1 or options.get().force_dp


def capturable():
    """Can this process group's collectives be recorded into a hipGraph?  RCCL kernels can (they are stream work); gloo
    moves data through the host.  LSPS_DP_GRAPHS=0 keeps data-parallel steps eager."""
    if not (dist.is_available() and dist.is_initialized()) or not options.get().dp_graphs:
        return False
    return dist.get_backend() == 'nccl'


def rank():
    return dist.get_rank() if dist.is_available() and dist.is_initialized() else 0


def cut_buckets(nbytes, bucket_bytes, segments=None):
    """Contiguous parameter index ranges [i0, i1) of the gradient arena, in READINESS order: backward produces the
    gradients of the last layers first, so the walk goes from the last parameter to the first and closes a bucket once
    it holds >= `bucket_bytes`; bucket 0 is therefore the tail of the arena and is complete first.  A tensor that alone
    reaches the bucket size travels alone (the discriminator's `model_S.3` weight, 75 MB, is ready a few launches into
    backward and must not wait for the small layers around it); no bucket crosses a segment start."""
    n = len(nbytes)
    starts = set(segments or ())
    out = []
    i1, acc = n, 0
    for i in range(n - 1, -1, -1):
        big = nbytes[i] >= bucket_bytes
        if big and i + 1 < i1:             # close what has been collected behind the big tensor
            out.append((i + 1, i1))
            i1, acc = i + 1, 0
        acc += nbytes[i]
        if big or acc >= bucket_bytes or i == 0 or i in starts:
            out.append((i, i1))
            i1, acc = i, 0
    return out


class GradReducer(object):
    """Bucketed, backward-overlapped all-reduce over a FlatArena's gradient buffer.

    Which gradients a backward pass will produce is LEARNED per step signature: the first time a signature (e.g.
    ``('dis_update', feat_mat, train_map)``) is seen, every bucket is launched from `finish()`; the set of parameters
    that were actually accumulated is recorded and, from the second step on, a bucket goes out the moment its last
    expected gradient has been accumulated, i.e. DURING backward.  Parameters that never receive a gradient (conv biases
    in front of an affine-free InstanceNorm, the `Post` head in `dis_update`, the `D` head in `post_update`, a frozen
    discriminator) therefore neither block a bucket nor need a hand-kept list.  All ranks run the same step, so they
    learn the same sets; a gradient that arrives for a bucket that has already gone out means the signature did not
    determine the graph and raises instead of reducing a half-filled bucket."""

    def __init__(self, arena, bucket_bytes=None, group=None, segments=None):
        """`segments`: start indices of the parameter groups that must never share a bucket (the generator arena holds
        `gen` then `map`: a step with train_map=False would otherwise all-reduce the idle Mapping's zeros)."""
        if bucket_bytes is None:
            bucket_bytes = options.get().bucket_bytes
        self.arena = arena
        self.group = group
        self.world = world()
        n = len(arena.params)
        self.buckets = cut_buckets([p.numel() * 4 for p in arena.params], bucket_bytes, segments)
        self.bucket_of = [0] * n
        for b, (a0, a1) in enumerate(self.buckets):
            for i in range(a0, a1):
                self.bucket_of[i] = b
        self._pending = None
        self._seen = {}
        self._works = []
        self._launched = None
        self._sig = None
        self._learned = {}
        self._scalars = None
        self.stats = dict(steps=0, buckets=0, early=0, exposed_ms=0.0, bytes=0)
        self.collect_stats = False         # bench.py: HIP events around the waits of finish() (never in a training run)
        self._exposed_events = []
        self.active = active()
        if self.active:
            arena.on_grad_ready = self._on_grad_ready

    # ---- per-backward protocol -------------------------------------------------------------
    def begin(self, sig=None, expected=None, scalars=None):
        """Call before the step's (first) backward.  `sig`: hashable step signature (see class doc); `expected`: explicit
        iterable of parameter indices that WILL get a gradient (overrides what was learned for `sig`).  `scalars`: a small
        device tensor (the step's loss scalars) summed over ranks by one tiny all-reduce that is launched here, ahead of the
        buckets, and hides under backward (`add_scalars` does the same later, for steps whose scalars only exist after a
        first partial backward); read it back with `reduced_scalars()` after `finish()`.
        A step may run SEVERAL backward passes between `begin` and `finish` (the estimate modes differentiate their two
        independent loss terms separately): a parameter's gradient is then accumulated — and its hook fires — once per pass
        that reaches it, so what is learned and counted down per parameter is the NUMBER of accumulations."""
        self._works = []
        self._launched = [False] * len(self.buckets)
        self._late = False
        self._sig = sig
        self._scalars = None
        self._pending = None
        self._seen = {}
        if not self.active:
            return
        if scalars is not None:
            self.add_scalars(scalars)
        if expected is None and sig is not None:
            expected = self._learned.get(sig)
        if expected is None:
            return
        self._expected = dict(expected) if isinstance(expected, dict) else dict((i, 1) for i in expected)
        self._pending = [0] * len(self.buckets)
        for i, cnt in self._expected.items():
            self._pending[self.bucket_of[i]] += cnt

    def add_scalars(self, scalars):
        if self.active and scalars is not None:
            self._scalars = scalars
            self._works.append(dist.all_reduce(scalars, op=dist.ReduceOp.SUM, group=self.group, async_op=True))

    def _on_grad_ready(self, i):
        self._seen[i] = self._seen.get(i, 0) + 1
        if self._pending is None:
            return
        b = self.bucket_of[i]
        if self._launched[b] or self._seen[i] > self._expected.get(i, 0):
            # raised in finish(): raising inside an autograd hook would be swallowed.  (An accumulation beyond the learned
            # count for a bucket still waiting is just as wrong: the bucket would leave one accumulation early.)
            if self._launched[b] or i in self._expected:
                self._late = True
            return
        self._pending[b] -= 1
        if self._pending[b] == 0:
            self._launch(b)

    def _launch(self, b):
        i0, i1 = self.buckets[b]
        self._launched[b] = True
        g = self.arena.grad_slice(i0, i1)
        self.stats['bytes'] += g.numel() * 4
        self._works.append(dist.all_reduce(g, op=dist.ReduceOp.SUM, group=self.group, async_op=True))

    def finish(self):
        """Call after backward, before the optimizer step: launches what is left, waits for all."""
        if not self.active:
            return
        if self._late:
            raise RuntimeError("GradReducer: a gradient arrived for a bucket that was already all-reduced (or more often "
                               "than learned); the step signature %r does not determine which parameters get gradients"
                               % (self._sig,))
        touched = self.arena.touched
        early = sum(self._launched)
        for b, (i0, i1) in enumerate(self.buckets):
            # ranks agree on which buckets carry gradients because they run the same step
            if not self._launched[b] and any(touched[i0:i1]):
                self._launch(b)
        # never inside a hipGraph capture: a captured event has no elapsed time (take_stats would raise)
        timed = self.collect_stats and self.arena.flat_g.is_cuda and not torch.cuda.is_current_stream_capturing()
        if timed:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
        for w in self._works:
            w.wait()
        if timed:
            e1.record()                        # the launch stream stalls between e0 and e1 = the exposed part
            self._exposed_events.append((e0, e1))
        if self._sig is not None:                  # accumulations per parameter in this step (hook calls)
            self._learned[self._sig] = dict((i, self._seen.get(i, 1)) for i, t in enumerate(touched) if t)
        self.stats['steps'] += 1
        self.stats['buckets'] += sum(self._launched)
        self.stats['early'] += early
        self.last_early, self.last_buckets = early, sum(self._launched)
        self._works = []
        self._pending = None

    def reduced_scalars(self):
        """The `scalars` tensor of `begin()` averaged over ranks (valid after `finish()`); None when inactive."""
        if self._scalars is None:
            return None
        return self._scalars / float(self.world)

    def take_stats(self):
        """Counters since the last call: steps, buckets all-reduced, buckets launched during backward, bytes, and the
        time the launch stream spent waiting for RCCL in `finish()` (HIP events around the waits)."""
        s = dict(self.stats)
        if self._exposed_events:
            torch.cuda.synchronize()
            s['exposed_ms'] = sum(a.elapsed_time(b) for a, b in self._exposed_events)
        self._exposed_events = []
        self.stats = dict(steps=0, buckets=0, early=0, exposed_ms=0.0, bytes=0)
        return s


def broadcast_from_rank0(tensors, group=None):
    """Replicas must start from the same state (weights AND Adam moments): rank 0's copy wins."""
    if not active():
        return
    for t in tensors:
        dist.broadcast(t, 0, group=group)


def local_device():
    """The HIP device this rank's collectives must stage on.  RCCL refuses two ranks on one device, and
    `torch.cuda.current_device()` is still 0 on every rank until somebody calls `set_device` — which the reference's
    call order does late (depth_train.py:103-107: `resume()` BEFORE `cuda(gpu)`), so a collective issued from `resume()`
    cannot rely on it: the launcher's LOCAL_RANK decides.  None for host-side backends (gloo)."""
    import os
    if not (dist.is_available() and dist.is_initialized()) or dist.get_backend() != 'nccl':
        return None
    return torch.device('cuda', int(os.environ.get('LOCAL_RANK', torch.cuda.current_device())))


def agree_from_rank0(value, device=None):
    """Rank 0's python value on every rank (e.g. the iteration count a resume() found); identity in a single process.
    `device`: where the pickled bytes are staged (default: `local_device()`, i.e. this rank's OWN GPU under RCCL even
    before the trainer has been moved there)."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return value
    box = [value]
    dist.broadcast_object_list(box, 0, device=device if device is not None else local_device())
    return box[0]


def drain_watchdog(timeout_s=5.0):
    """Blocks until RCCL's watchdog thread has retired every eager collective it tracks; True when that was CONFIRMED.

    Why: the watchdog polls the end events of the collectives on its work list (every ~100 ms); HIP refuses such a query
    (hipErrorCapturedEvent, which the watchdog turns into a process abort) once the process group's stream has joined a
    hipGraph capture.  Before a data-parallel step is captured the list must therefore be EMPTY, not just complete.  The
    list itself is not visible from Python, but the flight recorder is: the watchdog marks an entry `retired` at the moment
    it drops the finished work from its list (NOT the entry's `state`, which a dump refreshes by querying the events
    itself: 'completed' right after a device synchronise although the watchdog has not polled yet — r4k's bench under
    LSPS_FORCE_DP=1 aborted on exactly that).  Needs the recorder ON — it is off unless
    TORCH_FR_BUFFER_SIZE (torch < 2.9: TORCH_NCCL_TRACE_BUFFER_SIZE) is set before the process group is created
    (profiles/r4b_drain_probe.txt); bench.py and depth_train.py set it.  With it off, or on a torch without the call, the
    drain cannot be confirmed and the caller keeps the step eager."""
    import time
    if not (dist.is_available() and dist.is_initialized()) or dist.get_backend() != 'nccl':
        return True                        # no watchdog to race with
    torch.cuda.synchronize()               # every eager collective has finished on the device
    seen = _fr_entries(False)
    if not seen:                           # recorder off (or no such call): an empty 'active' list would prove nothing
        return False
    if 'retired' not in seen[-1]:          # a recorder without the field: nothing to confirm with
        return False
    t0 = time.time()
    while time.time() - t0 < timeout_s:
        ent = _fr_entries(False)
        if ent is not None and not [e for e in ent if not e.get('retired') and _fr_id(e) not in _captured_ids]:
            return True
        time.sleep(0.02)
    return False


# Collectives issued WHILE a hipGraph is captured are recorded by the flight recorder too, but the watchdog never tracks them
# (they only run in replays), so their entries stay 'scheduled' for ever: they must not count as pending eager work.
_captured_ids = set()


def _fr_entries(only_active):
    import pickle
    try:
        from torch._C._distributed_c10d import _dump_nccl_trace as dump
        return pickle.loads(dump(includeCollectives=True, includeStackTraces=False, onlyActive=only_active)).get('entries')
    except Exception:                      # noqa: BLE001  (no recorder in this build)
        return None


def _fr_id(e):
    return (e.get('pg_id'), e.get('record_id', e.get('collective_seq_id')), e.get('time_created_ns'))


def capture_begin():
    """Call right before a hipGraph capture of a data-parallel step; hand the result to `capture_end`."""
    if not capturable():
        return None
    return set(_fr_id(e) for e in (_fr_entries(False) or ()))


def capture_end(before):
    """Marks the flight-recorder entries born during the capture (see `_captured_ids`)."""
    if before is None:
        return
    for e in (_fr_entries(False) or ()):
        if _fr_id(e) not in before:
            _captured_ids.add(_fr_id(e))


def agree_all(flag, device=None):
    """True only if `flag` is true on EVERY rank (one tiny MIN all-reduce); identity in a single process."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return bool(flag)
    dev = device if device is not None else (local_device() or torch.device('cpu'))
    t = torch.tensor([1 if flag else 0], dtype=torch.int32, device=dev)
    dist.all_reduce(t, op=dist.ReduceOp.MIN)
    return bool(int(t.item()))


_oob_attempts = {}


def agree_all_oob(flag, timeout_s=60.0, tag='capture'):
    """True only if `flag` is true on EVERY rank, agreed through the process group's rendezvous STORE (TCPStore counters): no
    collective of the backend is issued, so nothing is added to RCCL's stream or its watchdog's work list.  This is what the
    capture decision of a data-parallel step needs: the ranks must take the same branch — a rank that stays eager
    while the others capture would, one call later, issue collectives the replaying ranks do not — and the agreement itself
    must not be an eager RCCL collective (it would have to be drained again).

    The verdict has ONE writer: every rank adds its vote, then the first rank that either sees all `n` votes or
    runs out of time publishes 'yes' / 'no' with `compare_set` on an empty verdict key, and EVERY rank — the publisher
    included — returns what that key holds.  A rank that arrives in the gap between another rank's last poll and its
    time-out therefore reads the same 'no' the timed-out rank wrote, instead of counting `n` votes for itself; and a verdict
    once written is never revised.  The key is derived from `tag` (the caller's graph signature) and a per-tag attempt
    counter, not from one process-wide call counter: ranks call this once per capture attempt OF THAT SIGNATURE in the same
    order, so one diverged or failed agreement cannot shift the keys of every later one.  Identity in a single process; a
    store error answers False on this rank (and, where the store still works, publishes 'no' for the others)."""
    import time
    import hashlib
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return bool(flag)
    tag_key = hashlib.sha1(repr(tag).encode()).hexdigest()[:16]
    attempt = _oob_attempts[tag_key] = _oob_attempts.get(tag_key, 0) + 1
    key = 'lsps_agree_%s_%d' % (tag_key, attempt)
    n = dist.get_world_size()
    store = None
    try:
        from torch.distributed.distributed_c10d import _get_default_store
        store = _get_default_store()
        store.add(key + '_ok', 1 if flag else 0)
        store.add(key + '_n', 1)
        t0 = time.time()
        while True:
            verdict = store.compare_set(key + '_verdict', '', '')       # read without writing ('' while nobody decided)
            if verdict:
                return verdict == b'yes'
            if store.add(key + '_n', 0) >= n:
                mine = 'yes' if store.add(key + '_ok', 0) == n else 'no'
            elif time.time() - t0 > timeout_s:
                mine = 'no'
            else:
                time.sleep(0.002)
                continue
            # single writer: only an EMPTY verdict is replaced; the answer is whatever the key holds afterwards
            return store.compare_set(key + '_verdict', '', mine) == b'yes'
    except Exception:                              # noqa: BLE001  (no store: nothing can be agreed => never capture)
        try:
            if store is not None:
                store.compare_set(key + '_verdict', '', 'no')
        except Exception:                          # noqa: BLE001
            pass
        return False


def barrier():
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        dist.barrier()


def all_reduce_mean_scalars(values, device):
    """Logging parity: average a small list of python floats over ranks (one tiny all-reduce)."""
    if not active():
        return values
    t = torch.tensor(values, dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return (t / world()).tolist()


def global_first(tensors, k=4):
    """The first `k` samples of the GLOBAL batch of every tensor in `tensors` (each this rank's contiguous shard along dim 0,
    equal shard sizes), identical on every rank — what `images[0:4]` means in the single-process reference
    (/root/reference/src/trainers/lsps_trainer.py:238) when the batch is sharded.  Shard >= k samples: they all live on rank 0,
    ONE broadcast of the concatenated slices (same trailing shape) or one per tensor.  Shard < k (e.g. global 16 over 8 ranks):
    the first ceil(k / shard) ranks hold them — one all-gather of the whole (small) shards, cut to k.  Identity when inactive."""
    if not active():
        return [t[0:k] for t in tensors]
    n = tensors[0].size(0)
    if n >= k:
        same = all(t.shape[1:] == tensors[0].shape[1:] and t.dtype == tensors[0].dtype for t in tensors)
        if same:
            first = torch.cat([t[0:k] for t in tensors], 0)
            dist.broadcast(first, 0)
            return list(first.split(k, 0))
        out = [t[0:k].contiguous() for t in tensors]
        for t in out:
            dist.broadcast(t, 0)
        return out
    out = []
    for t in tensors:
        parts = [torch.empty_like(t) for _ in range(world())]
        dist.all_gather(parts, t.contiguous())
        out.append(torch.cat(parts, 0)[0:k])
    return out


def shard_batch(t, dim=0):
    """This rank's contiguous shard of a global batch tensor."""
    w, r = world(), rank()
    n = t.size(dim)
    assert n % w == 0, "global batch must divide by world size"
    return t.narrow(dim, r * (n // w), n // w)
