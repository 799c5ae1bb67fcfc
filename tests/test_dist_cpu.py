"""N>1 path on CPU: world_size 2 and 8 over gloo.  The product's data-parallel machinery
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


WORLDS = (2, 8)       # 8 = BASELINE config 4's world size: the same code paths at the real rank count


def _spawn(target, world, args, n_results, timeout=300):
    """Runs `target(rank, world, port, *args, out)` in `world` spawned processes; returns `n_results` queue items."""
    ctx = mp.get_context('spawn')
    out = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=target, args=(r, world, port) + tuple(args) + (out,)) for r in range(world)]
    for p in procs:
        p.start()
    try:
        got = [out.get(timeout=timeout) for _ in range(n_results)]
    finally:
        for p in procs:
            p.join(timeout=90)
    for p in procs:
        assert p.exitcode == 0, [q.exitcode for q in procs]
    return got


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
    torch.set_num_threads(2 if world <= 2 else 1)
    from lsps_amd import dist as ldist
    from lsps_amd.optim import FlatArena
    try:
        hp = cases.hp_for('tiny')
        sds = cases.make_weights(hp, lsps_ref)
        N = max(4, world)                                   # world 8: one sample per domain and rank
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


@pytest.mark.parametrize('world', WORLDS)
def test_data_parallel_gradients_match_global_batch(world):
    res, = _spawn(_worker, world, (), 1)
    assert res['err'] < 1e-4, res
    assert res['nbuckets'] >= 3                       # several buckets at this bucket size
    assert sum(res['early']) >= res['nbuckets'] - 2   # buckets were launched DURING backward (overlap path)
    assert res['vals'] == [(world - 1) / 2.0, 2.0]    # mean over ranks of (rank, 2)
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


# ---------------------------------------------------------------------------------------------------------------
# round 2: learned step signatures, the product trainer's own `_step` bookkeeping, replicas made identical
# ---------------------------------------------------------------------------------------------------------------
class _NoStepOpt(object):
    def step(self):
        pass


def _product_step_worker(rank, world, port, out):
    """Drives the PRODUCT trainer's `_step` (lsps_amd/trainers/lsps_trainer.py) on CPU tensors over gloo: the trainer
    object is the real LSPSTrainer (constructor runs on CPU), its arenas/reducers are the real ones; only the losses
    are plain-torch stand-ins touching the same parameter subsets as the three update methods, and the Adam launch
    (HIP-only) is stubbed.  Nothing is passed as `expected`: the reducer has to learn it from the first step."""
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    os.environ['LSPS_BUCKET_BYTES'] = str(1 << 14)
    from lsps_amd import options
    options.reload_env()                    # the switches are read once per process, at import (lsps_amd/options.py)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    torch.set_num_threads(2 if world <= 2 else 1)
    try:
        from lsps_amd import dist as ldist
        from lsps_amd.optim import FlatArena
        import lsps_amd.trainers as prod
        torch.manual_seed(100 + rank)                       # every rank starts from DIFFERENT weights
        hp = cases.hp_for('tiny')
        tr = prod.LSPSTrainer(hp)
        tr.gpu = 'cpu-test'
        for key, opt, nets in (('dis', tr.dis_opt, (tr.dis,)), ('gen', tr.gen_opt, (tr.gen, tr.map))):
            arena = FlatArena(opt.param_groups[0]['params'])
            opt.arena = arena
            opt.flat_m, opt.flat_v = torch.zeros_like(arena.flat_p), torch.zeros_like(arena.flat_p)
            for p_ in opt.param_groups[0]['params']:
                opt.state[p_]['step'] = 3 + rank
            for n_ in nets:
                n_._arena = arena
            tr._reducers[key] = ldist.GradReducer(arena, segments=[0, len(list(nets[0].parameters()))])
        before = float(tr.dis_opt.arena.flat_p.abs().sum())
        tr.dis_opt.sync_from_rank0()
        tr.gen_opt.sync_from_rank0()
        sums = torch.tensor([float(tr.dis_opt.arena.flat_p.double().sum()), float(tr.gen_opt.arena.flat_p.double().sum())],
                            dtype=torch.float64)
        lo, hi = sums.clone(), sums.clone()
        dist.all_reduce(lo, op=dist.ReduceOp.MIN)
        dist.all_reduce(hi, op=dist.ReduceOp.MAX)
        steps = set(int(tr.dis_opt.state[p_]['step']) for p_ in tr.dis_opt.param_groups[0]['params'])

        named = dict(tr.dis.named_parameters())
        subsets = {   # which discriminator parameters each update method reaches (Post head / D head / trunk)
            ('dis_update', True, False): [k for k in named if not k.startswith('Post')],
            ('post_update', 0): [k for k in named if not k.startswith('model_D') and not k.startswith('model_B')],
            ('post_update', 3): [k for k in named if not k.startswith('model_D')],
        }
        report = {}
        for sig, keys in subsets.items():
            for it in range(2):
                tr.dis.zero_grad()
                loss = sum(((rank + 1.0) * named[k]).pow(2).sum() for k in keys)
                tr._step('dis', _NoStepOpt(), loss, ['dis_loss'], [loss.detach() * 0 + float(rank)], sig)
                red = tr._reducers['dis']
                report[(sig, it)] = (red.last_early, red.last_buckets, float(tr.dis_loss))
            # all-reduced gradient = sum over ranks of 2 (r+1)^2 p
            k0 = keys[0]
            want = sum(2.0 * (r + 1.0) ** 2 for r in range(world)) * named[k0].detach()
            report[(sig, 'err')] = float((named[k0].grad - want).abs().max())
        # gen arena: the residual-block biases never get a gradient -> they must not block any bucket
        gnamed = dict(tr.gen.named_parameters())
        live = [k for k in gnamed if not (k.endswith('.model.0.bias') or k.endswith('.model.3.bias'))]
        for it in range(2):
            tr.gen.zero_grad()
            loss = sum(gnamed[k].pow(2).sum() for k in live)
            tr._step('gen', _NoStepOpt(), loss, ['gen_total_loss'], [loss.detach()], ('gen_update', False, True))
            report[('gen', it)] = (tr._reducers['gen'].last_early, tr._reducers['gen'].last_buckets)
        if rank == 0:
            out.put(dict(report=report, same=bool(torch.equal(lo, hi)), steps=sorted(steps), moved=before))
    finally:
        dist.barrier()
        dist.destroy_process_group()


@pytest.mark.parametrize('world', WORLDS)
def test_product_step_learns_expected_gradients_and_overlaps_all_updates(world):
    res, = _spawn(_product_step_worker, world, (), 1)
    assert res['same'], "rank 0's weights were not broadcast"
    assert res['steps'] == [3], res['steps']                       # Adam step counts follow rank 0 too
    rep = res['report']
    for sig in (('dis_update', True, False), ('post_update', 0), ('post_update', 3)):
        e0, n0, _ = rep[(sig, 0)]
        e1, n1, scal = rep[(sig, 1)]
        assert n0 == n1 >= 3, (sig, n0, n1)
        assert e0 == 0, "first step of a signature: nothing known yet, everything goes at finish()"
        assert e1 >= n1 - 1, (sig, e1, n1)                         # learned: buckets go out DURING backward
        assert scal == (world - 1) / 2.0                            # scalars: mean over ranks of rank
        assert rep[(sig, 'err')] < 1e-5 * world
    assert rep[('gen', 0)][0] == 0 and rep[('gen', 1)][0] >= rep[('gen', 1)][1] - 1, rep


def test_late_gradient_for_a_launched_bucket_raises():
    """A signature that does not determine the graph must fail loudly, not reduce a half-filled bucket."""
    from lsps_amd import dist as ldist
    from lsps_amd.optim import FlatArena
    ps = [torch.nn.Parameter(torch.randn(8)), torch.nn.Parameter(torch.randn(8))]
    arena = FlatArena(ps)
    red = ldist.GradReducer(arena, bucket_bytes=1 << 20)
    red.active = True
    arena.on_grad_ready = red._on_grad_ready
    launched = []
    red._launch = lambda b: (launched.append(b), red._launched.__setitem__(b, True))
    arena.zero_grad()
    red.begin(('s',), expected=[0])
    arena.touched[0] = True
    red._on_grad_ready(0)                                    # the only expected gradient: the bucket goes out
    assert launched == [0] and not red._late
    arena.touched[1] = True
    red._on_grad_ready(1)                                    # ... and then one more gradient arrives for it
    assert red._late
    with pytest.raises(RuntimeError):
        red.finish()


def test_flat_adam_attach_keeps_moments_loaded_before_the_arena():
    """Driver order (depth_train.py:109-114): resume(load_opt=True) THEN cuda(): the loaded exp_avg / exp_avg_sq must
    survive attach() (they were overwritten by the zero arena views)."""
    from lsps_amd.optim import FlatAdam
    import lsps_amd.optim as lo
    ps = [torch.nn.Parameter(torch.randn(4, 3)), torch.nn.Parameter(torch.randn(5))]
    ref = torch.optim.Adam(ps, lr=1e-3)
    (ps[0].sum() * 2 + (ps[1] ** 2).sum()).backward()
    ref.step()
    ref.step()
    sd = ref.state_dict()
    opt = FlatAdam(ps, lr=1e-3)
    opt.load_state_dict(sd)
    m_before = [opt.state[p]['exp_avg'].clone() for p in ps]

    class _P(object):                                        # attach() insists on HIP parameters; fake the flag only
        def __init__(self, p):
            self.p = p
    orig = torch.Tensor.is_cuda
    try:
        lo.torch.Tensor.is_cuda = property(lambda self: True)
        orig_pin = torch.Tensor.pin_memory
        opt.attach()
    except Exception:
        raise
    finally:
        lo.torch.Tensor.is_cuda = orig
    for p, m in zip(ps, m_before):
        assert torch.equal(opt.state[p]['exp_avg'], m) and float(m.abs().sum()) > 0
        assert opt.state[p]['exp_avg'].data_ptr() >= opt.flat_m.data_ptr()
        assert int(opt.state[p]['step']) == 2


# ---------------------------------------------------------------------------------------------------------------
# round 3: the bucket layout of the FULL-WIDTH nets at the DEFAULT bucket size
# ---------------------------------------------------------------------------------------------------------------
def _full_width_reducer(tr, params, segments, monkey_log):
    """Real FlatArena + GradReducer (default LSPS_BUCKET_BYTES) over full-width oracle parameters; the collective itself
    is replaced by a recorder (single process), everything else — hooks, learned sets, launch order — is the product's."""
    from lsps_amd import dist as ldist
    from lsps_amd.optim import FlatArena
    arena = FlatArena(params)
    red = ldist.GradReducer(arena, segments=segments)
    red.active = True
    arena.on_grad_ready = red._on_grad_ready

    def launch(b):
        red._launched[b] = True
        i0, i1 = red.buckets[b]
        monkey_log.append((b, i0, i1, list(arena.touched), arena.grad_slice(i0, i1).numel() * 4))
    red._launch = launch
    return arena, red


def test_full_width_default_buckets_launch_in_readiness_order_and_skip_the_idle_mapping(monkeypatch):
    import yaml
    from lsps_amd import options
    monkeypatch.setattr(options, '_current', options.from_env({}))      # the DEFAULT bucket size
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    with open(os.path.join(repo, 'exps', 'nnyu.yaml')) as f:
        hp = yaml.safe_load(f)['train']['hyperparameters']
    assert not hp['train_map']
    torch.manual_seed(0)
    torch.set_num_threads(8)
    tr = lsps_ref.RefTrainer(hp, literal=True)
    with torch.no_grad():
        for net in (tr.gen, tr.dis, tr.vae, tr.map):
            for k, v in net.p.items():
                v.normal_(0, 0.02)
    T = torch.as_tensor
    b = cases.make_inputs(1)
    args = (T(b['xa']), T(b['la']), T(b['xb']), T(b['lb']), T(b['ca']), T(b['cb']))
    names = list(tr.dis.p)

    # ---- discriminator arena: dis_update twice (the first step of a signature learns, the second overlaps)
    log = []
    arena, red = _full_width_reducer(tr, tr.dis.parameters(), None, log)
    tr.dis.zero_grad = arena.zero_grad
    tr.dis_opt.step = lambda: None
    big = names.index('model_S.3.model.0.weight')
    assert (big, big + 1) in red.buckets, "a tensor >= the bucket size travels alone"
    assert red.buckets[0][1] == len(names), "bucket 0 = the tail of the arena (ready first)"
    for it in range(2):
        del log[:]
        sig = ('dis_update', True, False)
        red.begin(sig)
        tr.dis_update(args[0], args[1], args[2], args[3], args[4], args[5], hp)
        n_early = len(log)
        red.finish()
        if it == 0:
            assert n_early == 0
    early = log[:n_early]
    fronts = [i for i, k in enumerate(names) if k.startswith('model_A') or k.startswith('model_B')]
    sent = dict((b_, touched) for b_, i0, i1, touched, nb in early)
    bb = red.bucket_of[big]
    assert bb in sent, "the 75 MB trunk weight must go out during backward"
    assert not any(sent[bb][i] for i in fronts), "... before any front-end gradient exists"
    assert n_early >= len([1 for b_ in range(len(red.buckets)) if any(arena.touched[red.buckets[b_][0]:red.buckets[b_][1]])]) - 1
    # bytes on the wire = the discriminator arena minus nothing big: Post head is tiny and shares the head bucket
    assert sum(nb for _, _, _, _, nb in log) <= 4 * arena.total

    # ---- generator + Mapping arena: gen_update with train_map=False must not send a single Mapping byte
    log2 = []
    n_gen = len(tr.gen.parameters())
    arena2, red2 = _full_width_reducer(tr, tr.gen.parameters() + tr.map.parameters(), [0, n_gen], log2)
    tr.gen.zero_grad = arena2.zero_grad
    tr.gen_opt.step = lambda: None
    assert all(not (i0 < n_gen < i1) for i0, i1 in red2.buckets), "no bucket straddles the gen | map boundary"
    for it in range(2):
        del log2[:]
        red2.begin(('gen_update', False, True))
        tr.gen_update(args[0], args[1], args[2], args[3], hp)
        n_early2 = len(log2)
        red2.finish()
    assert all(i1 <= n_gen for _, i0, i1, _, _ in log2), "idle Mapping gradients (zeros) were all-reduced"
    gen_bytes = 4 * (arena2.offsets[n_gen])
    assert sum(nb for _, _, _, _, nb in log2) == gen_bytes
    assert n_early2 >= len(log2) - 1
    # what a pretrain step puts on the wire (MB): dis arena + gen part of the gen arena = 173.6 (was 242)
    total_mb = (sum(nb for _, _, _, _, nb in log) + gen_bytes) / 1e6
    assert abs(total_mb - 173.6) < 0.5, total_mb


# ---------------------------------------------------------------------------------------------------------------
# round 4: resume() under data parallelism runs a collective BEFORE cuda() (reference order,
# depth_train.py:103-107).  The count every rank continues from is rank 0's, whatever a rank found locally.
# ---------------------------------------------------------------------------------------------------------------
def _resume_worker(rank, world, port, root, out):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    torch.set_num_threads(1)
    try:
        import lsps_amd.trainers as prod
        from lsps_amd import dist as ldist
        assert ldist.local_device() is None                     # host-side backend: nothing to bind
        hp = cases.hp_for('tiny')
        torch.manual_seed(100 + rank)
        tr = prod.LSPSTrainer(hp)
        mine = os.path.join(root, 'rank%d' % rank)
        os.makedirs(mine, exist_ok=True)
        if rank == 0:                                           # only rank 0 holds a snapshot (no shared filesystem)
            torch.save(tr._dense_state(tr.gen), os.path.join(mine, 'pre_gen_%08d.pkl' % 7000))
            torch.save(tr._dense_state(tr.dis), os.path.join(mine, 'pre_dis_%08d.pkl' % 7000))
        elif rank == 1:
            # a HALF-WRITTEN file on this rank: torch.load raises RuntimeError (zip reader) / UnpicklingError, not OSError —
            # the rank must still enter the agreements and adopt rank 0's weights
            whole = os.path.join(mine, 'whole.pkl')
            torch.save(tr._dense_state(tr.gen), whole)
            blob = open(whole, 'rb').read()
            os.remove(whole)
            for name in ('pre_gen_%08d.pkl' % 7000, 'pre_dis_%08d.pkl' % 7000):
                with open(os.path.join(mine, name), 'wb') as f:
                    f.write(blob[:len(blob) // 3])
        dist.barrier()
        assert tr.gpu is None                                   # resume BEFORE cuda(), as the reference's driver calls it
        it = tr.resume(os.path.join(mine, 'pre'), idx=-1, load_opt=True)
        w = torch.cat([p.detach().reshape(-1).double() for p in tr.gen.parameters()]).sum().reshape(1)
        lo, hi = w.clone(), w.clone()
        dist.all_reduce(lo, op=dist.ReduceOp.MIN)
        dist.all_reduce(hi, op=dist.ReduceOp.MAX)
        same = bool(torch.equal(lo, hi))
        flags = (ldist.agree_all(True), ldist.agree_all(rank == 0), ldist.drain_watchdog())
        # the out-of-band agreement of the capture decision (rendezvous store, no collective of the backend: ADVICE r4)
        oob = (ldist.agree_all_oob(True), ldist.agree_all_oob(rank == 0), ldist.agree_all_oob(rank == 1), ldist.agree_all_oob(True))
        # a rank that arrives after another rank's time-out must read THAT rank's verdict, and the next agreement of
        # another signature must still rendezvous (per-tag attempt counters, single-writer verdict)
        import time
        if rank == world - 1:
            time.sleep(1.5)
        late = ldist.agree_all_oob(True, timeout_s=0.4, tag=('late',))
        dist.barrier()
        after = (ldist.agree_all_oob(True, tag=('late',)), ldist.agree_all_oob(True, tag=('other', 3)))
        # a CONFIGURATION error on a rank other than 0 (a state dict of another architecture): every rank raises, none hangs
        if rank == 1:
            torch.save({'encode_A.0.model.0.weight': torch.zeros(3, 3)}, os.path.join(mine, 'pre_gen_%08d.pkl' % 7000))
            torch.save(tr._dense_state(tr.dis), os.path.join(mine, 'pre_dis_%08d.pkl' % 7000))
        dist.barrier()
        try:
            tr.resume(os.path.join(mine, 'pre'), idx=-1)
            raised = None
        except RuntimeError as e:
            raised = type(e).__name__ + ':' + str(e)[:60]
        out.put((rank, it, flags, oob, same, late, after, raised))
    except Exception as e:                                       # the parent would otherwise wait for its timeout
        import traceback
        out.put((rank, 'error', repr(e) + traceback.format_exc()))
    finally:
        dist.barrier()
        dist.destroy_process_group()


@pytest.mark.parametrize('world', WORLDS)
def test_resume_before_cuda_agrees_on_rank0s_iteration_count(tmp_path, world):
    got = sorted(_spawn(_resume_worker, world, (str(tmp_path),), world, timeout=180))
    assert all(g[1] != 'error' for g in got), got
    assert [g[1] for g in got] == [7000] * world, got           # ranks without a (readable) snapshot still continue at 7000
    assert all(g[2] == (True, False, True) for g in got), got   # agree_all = AND over ranks; gloo has no watchdog to drain
    assert all(g[3] == (True, False, False, True) for g in got), got   # the same answer on every rank, round after round
    # (g[4]: the weights themselves are broadcast by cuda() -> sync_replicas(); before cuda() resume() only agrees on the count)
    assert len(set(g[5] for g in got)) == 1, got                # the late arrival and the timed-out ranks agree ...
    assert got[0][5] is False                                   # ... on the verdict of the rank that timed out
    assert all(g[6] == (True, True) for g in got), got          # and later agreements still rendezvous
    assert all(g[7] is not None for g in got), got              # the configuration error is raised on EVERY rank


def _global_first_worker(rank, world, port, out):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    torch.set_num_threads(1)
    try:
        from lsps_amd import dist as ldist
        res = {}
        # (global batch, what the case is): the first 4 on rank 0 only / spread over the first ranks / fewer than 4 in total
        for name, gb in (('rank0_only', 8 * world), ('exactly_4_per_rank', 4 * world), ('spread', 2 * world),
                         ('one_per_rank', world)):
            a = torch.arange(gb * 6, dtype=torch.float32).reshape(gb, 1, 2, 3)
            b = -torch.arange(gb * 6, dtype=torch.float32).reshape(gb, 1, 2, 3)
            fa, fb = ldist.global_first((ldist.shard_batch(a), ldist.shard_batch(b)), 4)
            k = min(4, gb)
            res[name] = bool(torch.equal(fa, a[:k]) and torch.equal(fb, b[:k]))
        # different trailing shapes (input_dim_a != input_dim_b): one collective per tensor
        a = torch.arange(8. * world).reshape(8 * world, 1)
        c = torch.arange(16. * world).reshape(8 * world, 2)
        fa, fc = ldist.global_first((ldist.shard_batch(a), ldist.shard_batch(c)), 4)
        res['mixed_shapes'] = bool(torch.equal(fa, a[:4]) and torch.equal(fc, c[:4]))
        out.put((rank, res))
    except Exception as e:
        import traceback
        out.put((rank, 'error: ' + repr(e) + traceback.format_exc()))
    finally:
        dist.barrier()
        dist.destroy_process_group()


@pytest.mark.parametrize('world', WORLDS)
def test_estimate_first4_are_the_global_first4_wherever_they_live(world):
    """post_update(mode >= 2) runs the generator on `images[0:4]` of the GLOBAL batch (reference lsps_trainer.py:238).  Sharded:
    rank 0 holds them when the shard has >= 4 samples (config 4: 128 per rank); with 16 samples over 8 ranks they are spread
    over ranks 0 and 1 — every rank must still see the same four."""
    got = _spawn(_global_first_worker, world, (), world, timeout=120)
    for rank, res in got:
        assert isinstance(res, dict) and all(res.values()), (rank, res)
    import inspect
    import lsps_amd.trainers.lsps_trainer as lt
    assert 'global_first((images_a, images_b), 4)' in inspect.getsource(lt.LSPSTrainer.post_update)


def test_global_first_single_process_is_a_plain_slice():
    from lsps_amd import dist as ldist
    a = torch.arange(12.).reshape(6, 2)
    fa, = ldist.global_first((a,), 4)
    assert torch.equal(fa, a[:4])


# ---------------------------------------------------------------------------------------------------------------
# round 4: a step may differentiate its loss terms in SEVERAL backward passes (post_update: the regression term first,
# beside the generator pass of the feature term).  The reducer learns the number of accumulations per parameter and
# launches a bucket only after the LAST one.
# ---------------------------------------------------------------------------------------------------------------
def _multi_pass_worker(rank, world, port, out):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    torch.set_num_threads(1)
    try:
        from lsps_amd import dist as ldist
        from lsps_amd.optim import FlatArena
        torch.manual_seed(5)
        ps = [torch.nn.Parameter(torch.randn(64, 64)) for _ in range(6)]
        arena = FlatArena(ps)
        red = ldist.GradReducer(arena, bucket_bytes=2 * 64 * 64 * 4)           # 3 buckets of 2 parameters
        assert len(red.buckets) == 3
        x = torch.full((64,), float(rank + 1))
        log = []
        orig = red._launch

        def launch(b):
            log.append(('launch', b, dict(red._seen)))
            orig(b)
        red._launch = launch
        res = []
        for it in range(3):
            arena.zero_grad()
            del log[:]
            red.begin(('two_terms',))
            term_a = sum((p @ x).sum() for p in ps[:4])                      # parameters 0..3
            term_b = sum((p @ x).pow(2).sum() for p in ps[2:])                # parameters 2..5  (2, 3 are reached twice)
            term_a.backward()
            launched_after_first = [e[1] for e in log]
            term_b.backward()
            early = sum(red._launched)
            red.finish()
            res.append(dict(learned=dict(red._learned[('two_terms',)]), after_first=launched_after_first, early=early,
                            order=[e[1] for e in log], g=arena.flat_g.clone()))
        # reference: one backward over the sum, all-reduced by hand
        for p in ps:
            p.grad = None
        (sum((p @ x).sum() for p in ps[:4]) + sum((p @ x).pow(2).sum() for p in ps[2:])).backward()
        ref = torch.cat([p.grad.reshape(-1) for p in ps])
        dist.all_reduce(ref)
        # two ranks: a + b is one rounding whatever the chunking; eight: the ring's summation order depends on the buffer cut
        same = torch.equal if world == 2 else (lambda a, b: torch.allclose(a, b, rtol=1e-5, atol=1e-4))
        out.put((rank, [dict(r, g=bool(same(r['g'], ref))) for r in res]))
    except Exception as e:
        import traceback
        out.put((rank, 'error', repr(e) + traceback.format_exc()))
    finally:
        dist.barrier()
        dist.destroy_process_group()


@pytest.mark.parametrize('world', WORLDS)
def test_reducer_counts_accumulations_of_a_multi_pass_step(world):
    got = sorted(_spawn(_multi_pass_worker, world, (), world, timeout=180))
    for rank, res in got:
        assert res != 'error', got
        first, second, third = res
        assert first['learned'] == {0: 1, 1: 1, 2: 2, 3: 2, 4: 1, 5: 1}
        assert first['early'] == 0 and first['g']                          # learning step: everything from finish()
        for r in (second, third):
            assert r['g'], "sum of the two passes, all-reduced"
            # bucket 0 = parameters 4, 5 (readiness order: the arena's tail), bucket 1 = 2, 3, bucket 2 = 0, 1
            assert r['after_first'] == [2], r                              # only the bucket whose members are done after pass 1
            assert r['early'] == 3 and sorted(r['order']) == [0, 1, 2], r   # the others DURING the second pass, none early
