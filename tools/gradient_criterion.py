#!/usr/bin/env python
"""Worst value of every gradient criterion of tests/golden/cases.py:compare over all golden step cases, for the CPU oracle
(`--impl oracle`, runs anywhere) or the HIP product path (`--impl hip`, GPU box).  The committed reports are
profiles/r4a_gradient_criterion_{oracle,hip}.txt ("report the measured worst in profiles/")."""
import argparse
import json
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, 'tests', 'golden'))

import numpy as np  # noqa: E402
import torch  # noqa: E402

import cases  # noqa: E402
from oracle import lsps_ref  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--impl', default='oracle', choices=['oracle', 'hip'])
    ap.add_argument('--sets', default='tiny,full,extra,expand,resx')
    args = ap.parse_args()
    if args.impl == 'hip':
        import lsps_amd.trainers as prod
        A = cases.NativeAdapter(prod, 'cuda')
    else:
        torch.set_num_threads(8)
        A = cases.NativeAdapter(lsps_ref, 'cpu')

    def G(c):
        return dict(np.load(os.path.join(REPO, 'tests', 'golden', 'golden_%s.npz' % c)))

    print("# %s vs the reference's golden vectors; value = worst over all gradient tensors of the set (relative to the "
          "tensor's abs-max; l2 relative to the norm)" % args.impl)
    print("# limits: max %.0e (resx %.0e) | %s | resx: %s" % (2e-2, 5e-2, cases.GRAD_ROBUST, cases.GRAD_ROBUST_RESX))
    for name in args.sets.split(','):
        kw, gr = {}, 2e-2
        if name == 'extra':
            R, g = cases.run_extra_cases(A, lsps_ref), G('extra')
        elif name == 'expand':
            R, g = cases.run_expand_cases(A, lsps_ref), G('expand')
        elif name == 'resx':
            R = cases.run_resx_cases(A, lsps_ref)
            g = {k: v for k, v in G('tiny').items() if k.split('/')[0] in R}
            gr, kw = 5e-2, dict(grad_robust=cases.GRAD_ROBUST_RESX)
        else:
            R = cases.run_step_cases(A, name, lsps_ref)
            g = {k: v for k, v in G(name).items() if k.split('/')[0] in R}
        rep = {}
        bad, worst = cases.compare(R, g, 1e-3, grad_rtol=gr, report=rep, **kw)
        ngrad = len(set(k.rsplit('/', 1)[0] for k in g if 'grads' in k.split('/')[0]))
        print("%-7s %4d gradient tensors, failures %d" % (name, ngrad, len(bad)))
        for k in ('grad_max', 'grad_q99', 'grad_q90', 'grad_one_minus_cosine', 'grad_l2', 'grad_mean', 'grad_absmax'):
            if k in rep:
                print("    %-22s %.3e   %s" % (k, rep[k][0], rep[k][1]))
        for b in bad[:10]:
            print("    FAIL", b)
    print(json.dumps({'impl': args.impl, 'torch': torch.__version__}))


if __name__ == '__main__':
    main()
