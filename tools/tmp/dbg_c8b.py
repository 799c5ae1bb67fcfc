import sys, os
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
import torch, torch.nn.functional as F
import test_c8_gpu as T
_lib, L, dev, st = T._env()
for rep in range(3):
  for (N, C, K) in ((5, 192, 16),):
    g = torch.Generator().manual_seed(N * 1000 + C + K + 1)
    w = T._rand(g, K, C, 3, 3, scale=1.0 / (3.0 * K ** 0.5))
    dy, add = T._rand(g, N, K, 32, 32), T._rand(g, N, C, 32, 32)
    o = F.leaky_relu(torch.randn(N, C, 32, 32, generator=g), 0.01).cuda()
    rs = (torch.rand(N * C, generator=g) + 0.5).cuda()
    dyc, oc = T._to_c8(dy), T._to_c8(o)
    ws, wsb = _lib.workspace(L.lsps_c8_conv3x3_workspace_bytes(C, K), dev)
    dx = torch.empty((N, C // 8, 32, 32, 8), dtype=T.BF, device=dev)
    dref = F.conv_transpose2d(T._rb(dy).double().cpu(), T._rb(w).double().cpu(), padding=1)
    _lib.check(L.lsps_c8_conv3x3_dgrad_inbwd(dyc.data_ptr(), w.data_ptr(), oc.data_ptr(), rs.data_ptr(), dx.data_ptr(), N, C, 32, 32, K, 0.01, ws, wsb, st), 'dinb')
    od = T._rb(o).double().cpu(); pos = od > 0
    gg, xh = torch.where(pos, dref, dref * 0.01), torch.where(pos, od, od / 0.01)
    m1, m2 = gg.mean((2, 3), keepdim=True), (gg * xh).mean((2, 3), keepdim=True)
    ref = rs.double().cpu().view(N, C, 1, 1) * (gg - m1 - xh * m2)
    got = T._from_c8(dx).double().cpu()
    err = (got - ref).abs().amax((2, 3)) / ref.abs().max()
    bad = torch.nonzero(err.flatten() > 5e-3).flatten()
    print(rep, (N, C, K), 'max', float(err.max()), 'bad (n,c):', [(int(i // C), int(i % C)) for i in bad[:12]], 'count', len(bad))
    for i in bad[:3]:
        n, c = int(i // C), int(i % C)
        d = (got[n, c] - ref[n, c])
        print('   plane', n, c, 'err max at', int(d.abs().argmax()), 'o there', float(od[n, c].flatten()[d.abs().argmax()]), 'got', float(got[n,c].flatten()[d.abs().argmax()]), 'ref', float(ref[n,c].flatten()[d.abs().argmax()]), 'n bad px', int((d.abs() > 5e-3 * ref.abs().max()).sum()))
