#!/usr/bin/env python
"""Times lsps_c8_stem_dgrad (csrc/c8stem.h) at the discriminator stem geometry (7x7 / stride 2, 128x128 images) for N = 64 / 256 / 512."""
import os, sys; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from lsps_amd import _lib
L=_lib.lib(); dev=torch.device('cuda'); st=_lib.stream(); BF=torch.bfloat16
for N in (64, 256, 512):
    H=128; K=64; P=64
    dy=torch.randn(N,8,P,P,8,device=dev).to(BF); y=torch.randn(N,8,P,P,8,device=dev).to(BF)
    w=torch.randn(K,1,7,7,device=dev)*0.1; dx=torch.empty(N,1,H,H,device=dev)
    f=lambda: _lib.check(L.lsps_c8_stem_dgrad(dy.data_ptr(),y.data_ptr(),w.data_ptr(),dx.data_ptr(),N,H,H,K,7,7,2,3,0.01,st),'d')
    f(); torch.cuda.synchronize()
    e0,e1=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): f()
    e1.record(); torch.cuda.synchronize()
    print(N, e0.elapsed_time(e1)/10, 'ms')
