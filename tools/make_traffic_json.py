#!/usr/bin/env python
"""Builds the per-kernel HBM-traffic files bench.py reads (`roofline.traffic`) from tools/pmc_pass.sh summaries:
  tools/make_traffic_json.py f32  gpurun_out/<dir>/pmc_summary.txt > profiles/r4_traffic.json        (tools/pmc_traffic.py)
  tools/make_traffic_json.py bf16 gpurun_out/<dir>/pmc_summary.txt > profiles/r4_traffic_bf16.json   (tools/pmc_c8.py)
Counters as MI355X_MICROARCH.md prescribes: separate passes for FETCH_SIZE and WRITE_SIZE (KB), calibrated on lsps_axpy over
3 x 1 GiB arrays in the same passes (2 GiB really read, 1 GiB really written): bytes = KB x 1024 x (known / counted)."""
import collections
import json
import os
import re
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def parse(path):
    ctr = collections.defaultdict(dict)
    kernel = None
    for line in open(path):
        if line.startswith('##') or not line.strip():
            continue
        m = re.match(r'\s+(\S+)\s+mean/dispatch\s+(\d+)\s+dispatches\s+(\d+)', line)
        if m and kernel:
            ctr[kernel][m.group(1)] = float(m.group(2))
        elif not line.startswith(' '):
            kernel = line.strip()
    return ctr


def find(ctr, pat):
    hits = [k for k in ctr if re.search(pat, k)]
    return ctr[hits[0]] if hits else None


def entry(c, cal, alg_bytes, gflop, note=None):
    fetch, write = c.get('FETCH_SIZE'), c.get('WRITE_SIZE')
    e = {'fetch_kb_raw': fetch, 'write_kb_raw': write,
         'hbm_bytes_per_launch': (fetch * cal['fetch'] + write * cal['write']) * 1024.0,
         'algorithmic_bytes_per_launch': alg_bytes, 'gflop_per_launch': gflop}
    if 'SQ_VALU_MFMA_BUSY_CYCLES' in c and 'GRBM_GUI_ACTIVE' in c:
        e['mfma_busy_frac'] = c['SQ_VALU_MFMA_BUSY_CYCLES'] / (1024.0 * c['GRBM_GUI_ACTIVE'] / 8.0)
    e['traffic_over_algorithmic'] = e['hbm_bytes_per_launch'] / alg_bytes
    if note:
        e['note'] = note
    return e


def main():
    kind, path = sys.argv[1], sys.argv[2]
    ctr = parse(path)
    ax = find(ctr, r'axpy_kernel')
    cal = {'fetch': 2097152.0 / ax['FETCH_SIZE'], 'write': 1048576.0 / ax['WRITE_SIZE']}
    from lsps_amd import _lib
    out = {'_csrc_sha16': _lib.csrc_sha16(),      # the sources these counters were taken on (bench.py checks it)
           '_what': 'HBM traffic per launch from rocprofv3 PMC on the build named by _csrc_sha16: separate --pmc FETCH_SIZE / --pmc WRITE_SIZE passes '
                    '(tools/pmc_pass.sh over tools/%s), built by tools/make_traffic_json.py from %s'
                    % ('pmc_traffic.py' if kind == 'f32' else 'pmc_c8.py', path),
           '_calibration': 'lsps_axpy on 3 x 1 GiB arrays in the same passes: FETCH_SIZE %d KB counted for 2 GiB read (x %.3f), '
                           'WRITE_SIZE %d KB for 1 GiB written (x %.3f)' % (ax['FETCH_SIZE'], cal['fetch'], ax['WRITE_SIZE'], cal['write'])}
    if kind == 'f32':
        N, C, K, HW = 256, 256, 256, 1024
        alg = (N * C * HW + N * K * HW) * 4 + K * C * 9 * 4
        gf = 2.0 * N * K * HW * C * 9 / 1e9
        out['shape'] = '3x3 256->256 @32x32, N=256 (%.2f algorithmic GFLOP per launch)' % gf
        out['wino4_f3x3_kernel'] = entry(find(ctr, r'wino4_f3x3_kernel'), cal, alg, gf,
                                         'an XCD keeps two of the eight 32-channel U slices in its L2, so four XCDs fetch every image')
        out['wino4_w3x3_kernel'] = entry(find(ctr, r'wino4_w3x3_kernel'), cal, alg, gf,
                                         'fetch = dy + x once; written = the 8 x 36 x 256 KB partial sums, read again by the reduce kernel')
    else:
        N, C, K, HW = 512, 256, 256, 1024
        act = N * C * HW * 2
        gf = 2.0 * N * K * HW * C * 9 / 1e9
        out['shape'] = '3x3 256->256 @32x32 on C8 bf16 tensors, N=512 (%.3f algorithmic GFLOP per launch)' % gf
        modes = {'1 conv+IN+LReLU': (r'c8_conv3x3_kernel<1>', 2 * act, 44), '2 conv+IN+residual': (r'c8_conv3x3_kernel<2>', 3 * act, 44),
                 '3 dgrad+IN backward': (r'c8_conv3x3_kernel<3>', 3 * act, 30), '0 dgrad+skip': (r'c8_conv3x3_kernel<0>', 3 * act, 30)}
        per, tot_w, hb, ab = {}, 0, 0.0, 0.0
        for name, (pat, alg, wgt) in modes.items():
            c = find(ctr, pat)
            if not c:
                continue
            e = entry(c, cal, alg + K * C * 9 * 2, gf)
            per[name] = e
            tot_w += wgt
            hb += wgt * e['hbm_bytes_per_launch']
            ab += wgt * e['algorithmic_bytes_per_launch']
        out['c8_conv3x3_kernel'] = {'hbm_bytes_per_launch': hb / tot_w, 'algorithmic_bytes_per_launch': ab / tot_w, 'gflop_per_launch': gf,
                                    'note': 'launch-weighted mean over the four epilogue modes in the mix of one pretrain step (44 / 44 / 30 / 30)',
                                    'per_mode': per}
        out['c8_wgrad_kernel'] = entry(find(ctr, r'c8_wgrad_kernel'), cal, 2 * act * 1.0 + K * C * 9 * 4, gf,
                                       'fetch = x + dy once; written = the partial sums, read again by c8_wgrad_reduce_kernel')
    print(json.dumps(out, indent=1))


if __name__ == '__main__':
    main()
