#!/usr/bin/env python
"""HBM roofline of the data-step kernels (lsps_crop_normalize / lsps_crop_augment): algorithmic bytes = one read +
one write of every crop (2 x N x 128 x 128 x 4 B), timed with HIP events on the launch stream.
usage: python tools/bench_data.py [--n 256,1024,4096] [--iters 20]"""
import argparse
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests', 'golden'))
import numpy as np  # noqa: E402
import torch  # noqa: E402

import data_cases  # noqa: E402
from lsps_amd import data as ldata  # noqa: E402

HBM_PEAK_GBS = 8000.0


def time_it(fn, iters):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--n', default='256,1024,4096')
    ap.add_argument('--iters', type=int, default=20)
    a = ap.parse_args()
    pipe = ldata.CropPipeline('cuda:0')
    base = [data_cases.make_sample(100 + k) for k in range(16)]
    rs = np.random.RandomState(1)
    res = []
    for N in [int(v) for v in a.n.split(',')]:
        dpt = torch.from_numpy(np.stack([base[k % 16]['dpt'] for k in range(N)])[:, None]).cuda()
        comz = np.array([base[k % 16]['com3D'][2] for k in range(N)], np.float32)
        cubez = np.full((N,), 300.0, np.float32)
        plans = []
        for k in range(N):
            s = base[k % 16]
            com2D = ldata.NYU_CAMERA.to_img(s['com3D'])
            M = np.asarray(ldata.crop_transform(ldata.NYU_CAMERA, com2D, s['cube'], (128, 128)), 'float32')
            plans.append(ldata.plan_augmentation(ldata.NYU_CAMERA, s['gt3D'], com2D, s['cube'], M, ['none', 'com', 'rot'], rs))
        table = torch.from_numpy(np.stack([p.prm for p in plans])).cuda()
        cz, hf = torch.from_numpy(comz).cuda(), torch.from_numpy(cubez / 2).cuda()
        x, y = torch.empty_like(dpt), torch.empty_like(dpt)
        L, lib = pipe.L, pipe._lib
        st = lib.stream()
        t_n = time_it(lambda: lib.check(L.lsps_crop_normalize(dpt.data_ptr(), cz.data_ptr(), hf.data_ptr(), x.data_ptr(), N,
                                                              128 * 128, st), 'n'), a.iters)
        t_a = time_it(lambda: lib.check(L.lsps_crop_augment(x.data_ptr(), table.data_ptr(), y.data_ptr(), N, 128, 128, st),
                                        'a'), a.iters)
        nbytes = 2.0 * N * 128 * 128 * 4
        for name, t in (('crop_normalize', t_n), ('crop_augment', t_a)):
            r = {'kernel': name, 'n_crops': N, 'ms': t, 'algorithmic_bytes': nbytes, 'GB/s': nbytes / t / 1e6,
                 'frac_of_hbm_peak': nbytes / t / 1e6 / HBM_PEAK_GBS, 'crops_per_s': N / t * 1e3}
            res.append(r)
            print(json.dumps(r))
    return res


if __name__ == '__main__':
    main()
