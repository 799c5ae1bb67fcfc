"""N>1 path on CPU: world_size 2 over gloo.  The product's data-parallel machinery
(lsps_amd.optim.FlatArena + lsps_amd.dist.GradReducer) is backend-agnostic; here it is driven with the
CPU oracle as the compute (tests may use the oracle) and checked against a single-process run on the
global batch: averaged shard gradients == global-batch gradients (InstanceNorm has no batch statistics,
all losses are batch means — SURVEY.md §8(e))."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import cases
from oracle import lsps_ref


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _dis_grads(hp, sds, batch, noise, arena_cls=None, reducer_cls=None, bucket_bytes=1 << 16):
    """One oracle dis_update backward (optimizer step suppressed); returns (flat grad, launched-early flags)."""
    tr = lsps_ref.RefTrainer(hp, literal=False)
    for net in ('gen', 'dis', 'vae', 'map'):
        getattr(tr, net).load_state_dict(sds[net])
    tr.dis_opt.step = lambda: None
    T = torch.as_tensor
    args = (T(batch['xa']), T(batch['la']), T(batch['xb']), T(batch['lb']), T(batch['ca']), T(batch['cb']), hp)
    if arena_cls is None:
        tr.dis_update(*args, noise=T(noise))
        return torch.cat([p.grad.reshape(-1) if p.grad is not None else torch.zeros(p.numel()) for p in tr.dis.parameters()]), None
    arena = arena_cls(tr.dis.parameters())
    tr.dis.zero_grad = arena.zero_grad
    red = reducer_cls(arena, bucket_bytes=bucket_bytes)
    used = [i for i, k in enumerate(tr.dis.p) if not k.startswith('Post')]     # dis_update never touches Post
    early = {}
    orig_finish = red.finish

    def finish():
        early['flags'] = list(red._launched)
        orig_finish()
    red.begin(expected=used)
    # dis_update zeroes grads first (arena.zero_grad), then backward fires the hooks
    tr.dis_update(*args, noise=T(noise))
    finish()
    flat = torch.cat([arena.flat_g[o:o + p.numel()] for p, o in zip(arena.params, arena.offsets)])
    return flat / dist.get_world_size(), (early['flags'], len(red.buckets), list(arena.touched))


def _worker(rank, world, port, out):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    torch.set_num_threads(2)
    from lsps_amd import dist as ldist
    from lsps_amd.optim import FlatArena
    try:
        hp = cases.hp_for('tiny')
        sds = cases.make_weights(hp, lsps_ref)
        N = 4
        b = cases.make_inputs(N)
        lat = cases.latent_shape(hp, 2 * N)
        nz = cases.noise(lat, 99)
        per = N // world
        sl = slice(rank * per, (rank + 1) * per)
        shard = {k: v[sl] for k, v in b.items()}
        assert np.array_equal(ldist.shard_batch(torch.as_tensor(b['xa'])).numpy(), shard['xa'])
        nz_shard = np.concatenate([nz[sl], nz[N + rank * per:N + (rank + 1) * per]], 0)
        g_dp, (early, nbuckets, touched) = _dis_grads(hp, sds, shard, nz_shard, FlatArena, ldist.GradReducer)
        vals = ldist.all_reduce_mean_scalars([float(rank), 2.0], 'cpu')
        if rank == 0:
            g_ref, _ = _dis_grads(hp, sds, b, nz)
            # FlatArena pads every tensor to 4 elements: compare tensor by tensor
            out.put(dict(err=float((g_dp - g_ref).abs().max() / g_ref.abs().max()), early=early, nbuckets=nbuckets,
                         touched=touched, vals=vals))
    finally:
        dist.barrier()
        dist.destroy_process_group()


def test_data_parallel_gradients_match_global_batch():
    ctx = mp.get_context('spawn')
    out = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, out)) for r in range(2)]
    for p in procs:
        p.start()
    res = out.get(timeout=300)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert res['err'] < 1e-4, res
    assert res['nbuckets'] >= 3                       # several buckets at this bucket size
    assert sum(res['early']) >= res['nbuckets'] - 2   # buckets were launched DURING backward (overlap path)
    assert res['vals'] == [0.5, 2.0]
    assert not all(res['touched'])                    # Post head untouched -> skipped by Adam, not all-reduced


def test_reducer_is_a_noop_single_process():
    from lsps_amd import dist as ldist
    from lsps_amd.optim import FlatArena
    ps = [torch.nn.Parameter(torch.randn(5, 3)), torch.nn.Parameter(torch.randn(7))]
    arena = FlatArena(ps)
    red = ldist.GradReducer(arena)
    arena.zero_grad()
    red.begin()
    (ps[0].sum() * 2 + ps[1].sum() * 3).backward()
    red.finish()
    assert arena.touched == [True, True]
    assert torch.allclose(arena.flat_g[:15], torch.full((15,), 2.0))
    assert ps[0].data.data_ptr() == arena.flat_p.data_ptr()
    assert ldist.world() == 1 and ldist.rank() == 0
