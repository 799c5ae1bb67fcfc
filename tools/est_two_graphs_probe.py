#!/usr/bin/env python
"""Feasibility probe for the estimate3 step as TWO single-stream hipGraphs replayed side by side (instead of one two-branch
graph whose second branch hipGraphLaunch starts ~2.5 ms late): G1 = regression forward + backward on the launch stream,
G2 = generator pass + dis.feats forward + its gradients (torch.autograd.grad, no AccumulateGrad) on a second stream, then
join + gradient adds + Adam.  Prints ms per step of the pair against the trainer's own graphed post_update."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import bench  # noqa: E402
from lsps_amd import ops  # noqa: E402
import lsps_amd.trainers as trainers  # noqa: E402

hp = bench.load_hp()
tr = trainers.LSPSTrainer(hp)
tr.cuda(0)
dev = torch.device('cuda', 0)
b = bench.make_device_batch(128, dev)
tr.gen.train(); tr.dis.train()
main = torch.cuda.current_stream()
side = torch.cuda.Stream()
params = [p for p in tr.dis.parameters()]


def part_a():                       # regression branch: forward + backward into the arena
    _, pred, _ = tr.dis.regress_a(b['xa'])
    with torch.no_grad():
        target, _, _ = tr.vae.encode(b['la'])
    loss = hp['reg_w'] * ops.l2_loss(pred, target.reshape(pred.shape))
    loss.backward()
    return loss.detach()


def part_b():                       # feature branch: no AccumulateGrad, gradients as tensors
    with torch.no_grad():
        x_aa, x_ba, x_ab, x_bb, _ = tr.gen(b['xa'][0:4], b['xb'][0:4])
    f_aa, f_ba, f_ab, f_bb = tr.dis.feats(x_aa, x_ba, x_ab, x_bb)
    loss = hp['feature_w_reg'] * (ops.l1_loss(f_ab, f_aa) + ops.l1_loss(f_ba, f_bb))
    used = [p for p in params]
    grads = torch.autograd.grad(loss, used, allow_unused=True)
    return loss.detach(), grads


def eager_step():
    tr.dis.zero_grad()
    tr._declare_frozen('post_update')
    ops.weight_cache_begin(dev)
    try:
        la = part_a()
        side.wait_stream(main)
        with torch.cuda.stream(side):
            lb, grads = part_b()
        main.wait_stream(side)
    finally:
        ops.weight_cache_end()
    return la, lb, grads


for _ in range(3):
    eager_step()
torch.cuda.synchronize()
# capture: each part alone, own pool, own stream
ga, gb = torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph()
tr.dis.zero_grad()
tr._declare_frozen('post_update')
ops.weight_cache_begin(dev)
with torch.cuda.graph(ga):
    la = part_a()
ops.weight_cache_end()
torch.cuda.synchronize()
tr._declare_frozen('post_update')
ops.weight_cache_begin(dev)
with torch.cuda.graph(gb, stream=side):
    lb, grads = part_b()
ops.weight_cache_end()
torch.cuda.synchronize()
idx = [i for i, g in enumerate(grads) if g is not None]
dst = [params[i].grad for i in idx]
src = [grads[i] for i in idx]
print("captured: %d gradient tensors from the feature branch" % len(idx))


def two_graph_step(order):
    tr.dis.zero_grad()
    side.wait_stream(main)
    if order == 'a_first':
        ga.replay()
        with torch.cuda.stream(side):
            gb.replay()
    else:
        with torch.cuda.stream(side):
            gb.replay()
        ga.replay()
    main.wait_stream(side)
    torch._foreach_add_(dst, src)
    for i in range(len(tr.dis_opt.arena.touched)):
        tr.dis_opt.arena.touched[i] = True
    tr.dis_opt.step()


def timed(fn, k=100):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(k):
        fn()
    torch.cuda.synchronize()
    return 1e3 * (time.perf_counter() - t0) / k


def host_cost():
    torch.cuda.synchronize()
    t0 = time.perf_counter(); ga.replay(); t1 = time.perf_counter()
    with torch.cuda.stream(side):
        gb.replay()
    t2 = time.perf_counter()
    torch.cuda.synchronize()
    t3 = time.perf_counter()
    return 1e3 * (t1 - t0), 1e3 * (t2 - t1), 1e3 * (t3 - t0)


for _ in range(3):
    host_cost()
print("host time of ga.replay() %.3f ms, gb.replay() %.3f ms, both finished after %.3f ms" % host_cost())
import threading


def threaded_step():
    tr.dis.zero_grad()
    side.wait_stream(main)
    torch.cuda.synchronize()

    def run_b():
        with torch.cuda.stream(side):
            gb.replay()
    th = threading.Thread(target=run_b)
    t0 = time.perf_counter()
    th.start()
    ga.replay()
    th.join()
    main.wait_stream(side)
    torch.cuda.synchronize()
    return 1e3 * (time.perf_counter() - t0)


for _ in range(3):
    threaded_step()
print("two graphs replayed from two host threads: both finished after %.3f ms" % min(threaded_step() for _ in range(10)))
print("two graphs, regression first: %.3f ms" % timed(lambda: two_graph_step('a_first')))
print("two graphs, feature first:    %.3f ms" % timed(lambda: two_graph_step('b_first')))
print("eager parts (same structure): %.3f ms" % timed(lambda: (eager_step(), tr.dis_opt.step())))
tr.use_graphs(True)
step = lambda: tr.post_update(b['xa'], b['la'], b['xb'], b['lb'], b['ca'], b['cb'], 3, hp)   # noqa: E731
print("trainer post_update, one graph: %.3f ms" % timed(step))
