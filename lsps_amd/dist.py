"""Data-parallel gradient exchange for the depth path (NEW: the reference is single-GPU).

One process per GPU (`torch.distributed`, backend "nccl" = RCCL over xGMI).  The path shards
naturally along the batch: InstanceNorm has no batch statistics and every loss is a batch mean,
so rank-local mean-loss gradients averaged over ranks ARE the global-batch gradients
(SURVEY.md §8(e)).  The only exchange step is one sum-all-reduce of the gradient arena per
optimizer step; 1/world is folded into the Adam kernel (`FlatAdam.grad_scale`).

xGMI is point-to-point (7 links x ~153 GB/s per GPU), so the arena is cut into a FEW LARGE
contiguous buckets (default 32 MiB) — per-link-bound ring steps want big messages — ordered so
that the bucket whose gradients are produced FIRST by backward (the last layers) is launched
first, on RCCL's own stream, while backward keeps producing the rest (overlap).  Buckets are
slices of the flat gradient buffer: no flatten / unflatten copies.

The reducer is backend-agnostic (works on gloo/CPU tensors), which is how the N>1 path is
covered by CPU tests (tests/test_dist_cpu.py, world_size 2).
"""
import torch
import torch.distributed as dist


def world():
    return dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1


def active():
    """True when gradients must be exchanged.  LSPS_FORCE_DP=1 also runs the exchange in a 1-rank process
    group (used to smoke-test the RCCL call pattern on a single-GPU box)."""
    import os
    if not (dist.is_available() and dist.is_initialized()):
        return False
    return dist.get_world_size() > 1 or os.environ.get('LSPS_FORCE_DP') == '1'


def rank():
    return dist.get_rank() if dist.is_available() and dist.is_initialized() else 0


class GradReducer(object):
    """Bucketed, backward-overlapped all-reduce over a FlatArena's gradient buffer."""

    def __init__(self, arena, bucket_bytes=32 << 20, group=None):
        self.arena = arena
        self.group = group
        self.world = world()
        n = len(arena.params)
        # buckets = contiguous parameter index ranges [i0, i1), cut at >= bucket_bytes
        self.buckets = []
        i0, acc = 0, 0
        for i, p in enumerate(arena.params):
            acc += p.numel() * 4
            if acc >= bucket_bytes or i == n - 1:
                self.buckets.append((i0, i + 1))
                i0, acc = i + 1, 0
        self.bucket_of = [0] * n
        for b, (a0, a1) in enumerate(self.buckets):
            for i in range(a0, a1):
                self.bucket_of[i] = b
        self._pending = None
        self._works = []
        self._launched = None
        self.active = active()
        if self.active:
            arena.on_grad_ready = self._on_grad_ready

    # ---- per-backward protocol -------------------------------------------------------------
    def begin(self, expected=None):
        """Call before backward.  `expected`: iterable of parameter indices that WILL get a gradient
        (same on every rank).  With it, a bucket is launched as soon as its last expected gradient has
        been accumulated (overlap with the rest of backward); without it, everything goes at finish()."""
        self._works = []
        self._launched = [False] * len(self.buckets)
        if expected is None or not self.active:
            self._pending = None
            return
        self._pending = [0] * len(self.buckets)
        for i in expected:
            self._pending[self.bucket_of[i]] += 1

    def _on_grad_ready(self, i):
        if self._pending is None:
            return
        b = self.bucket_of[i]
        self._pending[b] -= 1
        if self._pending[b] == 0 and not self._launched[b]:
            self._launch(b)

    def _launch(self, b):
        i0, i1 = self.buckets[b]
        self._launched[b] = True
        self._works.append(dist.all_reduce(self.arena.grad_slice(i0, i1), op=dist.ReduceOp.SUM,
                                           group=self.group, async_op=True))

    def finish(self):
        """Call after backward, before the optimizer step: launches what is left, waits for all."""
        if not self.active:
            return
        touched = self.arena.touched
        for b, (i0, i1) in enumerate(self.buckets):
            # ranks agree on which buckets carry gradients because they run the same step
            if not self._launched[b] and any(touched[i0:i1]):
                self._launch(b)
        for w in self._works:
            w.wait()
        self._works = []
        self._pending = None


def all_reduce_mean_scalars(values, device):
    """Logging parity: average a small list of python floats over ranks (one tiny all-reduce)."""
    if not active():
        return values
    t = torch.tensor(values, dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return (t / world()).tolist()


def shard_batch(t, dim=0):
    """This rank's contiguous shard of a global batch tensor."""
    w, r = world(), rank()
    n = t.size(dim)
    assert n % w == 0, "global batch must divide by world size"
    return t.narrow(dim, r * (n // w), n // w)
