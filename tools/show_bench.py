#!/usr/bin/env python
"""Prints the interesting parts of bench.py JSON lines (files given on the command line)."""
import json
import sys
for f in sys.argv[1:]:
    try:
        j = json.loads([l for l in open(f) if l.startswith('{')][-1])
    except Exception as e:
        print(f, 'ERR', e)
        print(open(f).read()[-1500:])
        continue
    print(f, 'value', round(j['value'], 3), 'ms', round(j['ms_per_step'], 1), 'n_ranks', j.get('n_ranks'))
    if j.get('data_parallel'):
        print('  dp', json.dumps(j['data_parallel']))
    if j.get('cpu_baseline'):
        print('  cpu', {k: v for k, v in j['cpu_baseline'].items() if k not in ('sample',)})
        print('  speedups', j.get('speedup_vs_cpu_baseline'), j.get('estimate3_speedup_vs_cpu_baseline'))
    for k, v in (j.get('other_workloads') or {}).items():
        print('  extra', k, {a: (round(b, 3) if isinstance(b, float) else b) for a, b in v.items() if a != 'note'})
    r = j['roofline']
    print('  roof', r['kernel'][:40], round(r['achieved'], 1), round(r['frac'], 3), r.get('mfma_issued_frac'), 'traffic', r.get('traffic'))
    tot = 0
    for k, v in sorted(r['per_kernel'].items(), key=lambda kv: -kv[1]['total_ms']):
        tot += v['total_ms']
        print('     %-28s calls %4d  %8.1f ms/step  avg %.3f  %.1f TF' % (k, v['calls'], v['total_ms'] / r.get('per_kernel_steps', j['steps']), v['avg_ms'], v['tflops']))
    print('     sum of conv spans per step %.1f ms' % (tot / r.get('per_kernel_steps', j['steps'])))
