import sys
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests'); sys.path.insert(0, '/root/repo/hrnet-hand-pose-estimation_amd/lib')
import numpy as np, torch, torch.nn.functional as F
import hip_helpers as hh
from hipnet import _capi as C
dtype = torch.float32
N, H, W, Cc = 2, 16, 16, 480
g = torch.Generator().manual_seed(1)
y32 = (torch.randn(N, Cc, H, W, generator=g) * (torch.rand(Cc, generator=g).view(1,-1,1,1)+0.2) + torch.randn(Cc, generator=g).view(1,-1,1,1)*2)
gam = torch.rand(Cc, generator=g) + 0.5; bet = torch.rand(Cc, generator=g) - 0.5
gout = torch.randn(N, Cc, H, W, generator=g)
def ref(dt):
    y = y32.to(dt).requires_grad_(True); ga = gam.to(dt).requires_grad_(True); be = bet.to(dt).requires_grad_(True)
    z = F.batch_norm(y, None, None, ga, be, True, 0.1, 1e-5)
    out = F.relu(z); out.backward(gout.to(dt))
    return y.grad.double(), ga.grad.double(), be.grad.double()
dy64, dg64, db64 = ref(torch.float64)
dy32, dg32, db32 = ref(torch.float32)
d = hh.DEV
mean = y32.double().mean((0,2,3)); var = y32.double().var((0,2,3), unbiased=False); invstd = (1/torch.sqrt(var+1e-5))
mean_d, invstd_d, gam_d = mean.float().to(d), invstd.float().to(d), gam.to(d)
scale = (gam.double()*invstd).float().to(d); shift = (bet.double() - mean*gam.double()*invstd).float().to(d)
yd, gd = hh.nhwc(y32, dtype), hh.nhwc(gout, dtype)
blocks = C.call('hrnet_reduce_blocks', N, H, W, Cc)
part = torch.empty(blocks, 2, Cc, device=d)
C.call('hrnet_bn_bwd_reduce', 0, part.data_ptr(), gd.data_ptr(), None, yd.data_ptr(), scale.data_ptr(), shift.data_ptr(), N, H, W, Cc, 0, 1, C.stream_ptr())
dgam, dbet, coef = torch.zeros(Cc, device=d), torch.zeros(Cc, device=d), torch.empty(3*Cc, device=d)
C.call('hrnet_bn_bwd_finalize', part.data_ptr(), blocks, Cc, float(N*H*W), gam_d.data_ptr(), mean_d.data_ptr(), invstd_d.data_ptr(), dgam.data_ptr(), dbet.data_ptr(), coef.data_ptr(), 0, C.stream_ptr())
dy = torch.empty(N, H, W, Cc, dtype=dtype, device=d)
C.call('hrnet_grad_term', 0, dy.data_ptr(), gd.data_ptr(), None, yd.data_ptr(), scale.data_ptr(), shift.data_ptr(), coef.data_ptr(), N, H, W, Cc, 0, 1, 0, C.stream_ptr())
def e(a, b): return ((a.double()-b).abs().max()/b.abs().max()).item()
print('dgamma hip %.2e torch32 %.2e' % (e(dgam.cpu(), dg64), e(dg32, dg64)))
print('dbeta  hip %.2e torch32 %.2e' % (e(dbet.cpu(), db64), e(db32, db64)))
print('dy     hip %.2e torch32 %.2e' % (e(hh.from_nhwc(dy), dy64), e(dy32, dy64)))
