import sys, torch
sys.path.insert(0, '/root/repo')
import torch.nn.functional as F
from lsps_amd import ops
def rel(a,b):
    a,b=a.detach().cpu().double(),b.detach().cpu().double(); return float((a-b).abs().max()/b.abs().max())
g=torch.Generator().manual_seed(0)
N,C,H,K=3,256,32,256
x=(torch.rand(N,C,H,H,generator=g)*2-1).requires_grad_(True); w=((torch.rand(K,C,3,3,generator=g)*2-1)*0.1).requires_grad_(True)
xr=x.detach().double().requires_grad_(True); wr=w.detach().double().requires_grad_(True)
yr=F.conv2d(xr,wr,None,1,1); gy=torch.rand(*yr.shape,generator=g)*2-1; yr.backward(gy.double())
for mode in ('f32','f32_split','bf16'):
    ops.set_math_mode(mode)
    xd,wd=x.detach().cuda().requires_grad_(True), w.detach().cuda().requires_grad_(True)
    y=ops.conv2d(xd,wd,None,1,1); y.backward(gy.cuda()); torch.cuda.synchronize()
    print(mode, 'y %.2e dx %.2e dw %.2e' % (rel(y,yr), rel(xd.grad,xr.grad), rel(wd.grad,wr.grad)))
ops.set_math_mode('f32')
