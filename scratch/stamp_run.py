import sys, os
os.environ['HRNET_HIP_LIB'] = '/root/repo/scratch/libstamp.so'
sys.path.insert(0, '/root/repo/tests'); sys.path.insert(0, '/root/repo/hrnet-hand-pose-estimation_amd/lib')
import torch, numpy as np
import hip_helpers as hh
from hipnet import _capi as C
dt = torch.bfloat16
N, H, Cin = 64, int(sys.argv[1]), int(sys.argv[2]); Cout = Cin; ks = int(sys.argv[4]) if len(sys.argv) > 4 else 3
aff = int(sys.argv[3]) if len(sys.argv) > 3 else 1
x = torch.randn(N, H, H, Cin, device='cuda').to(dt)
w = torch.randn(Cout, Cin, ks, ks) * 0.05
wp, cop, cip = hh.pack_weights(w, dt)
sc = torch.rand(Cin, device='cuda') + 0.5; sh = torch.rand(Cin, device='cuda') - 0.5
y = torch.empty(N, H, H, cop, device='cuda', dtype=dt)
gx = C.call('hrnet_conv_tiles', N, H, H, cop, ks, 1)
buf = torch.zeros(gx * 8 * 32, dtype=torch.int64, device='cuda')
def run():
    C.call('hrnet_conv2d', 1, x.data_ptr(), wp.data_ptr(), sc.data_ptr() if aff else None, sh.data_ptr() if aff else None, None, y.data_ptr(), buf.data_ptr(), N, H, H, Cin, H, H, cop, ks, 1, 0, aff, 0, C.stream_ptr())
for _ in range(3): run()
buf.zero_(); torch.cuda.synchronize(); run(); torch.cuda.synchronize()
st = buf.cpu().numpy().reshape(-1, 32)
st = st[st[:, 0] > 0]
nst = (st > 0).sum(1)
print('workgroups', len(st), 'stamps per WG', np.bincount(nst))
t0 = st[:, 0].min()
rel = (st - t0).astype(np.float64)
rel[st == 0] = np.nan
start = rel[:, 0]
print('WG start spread (cycles): min %.0f median %.0f max %.0f' % (np.nanmin(start), np.nanmedian(start), np.nanmax(start)))
last = np.array([r[n-1] for r, n in zip(rel, nst)])
print('WG end: median %.0f max %.0f  ; WG lifetime median %.0f' % (np.nanmedian(last), np.nanmax(last), np.nanmedian(last - start)))
names = ['issue load0'] 
d = np.diff(rel, axis=1)
labels = ['load0 issue'] + sum([['store(wait+xform) s%d' % s, 'barrier s%d' % s, 'issue next + MFMA s%d' % s, 'epilogue s%d' % s, 'barrier2 s%d' % s] for s in range(6)], [])
for i in range(min(d.shape[1], 16)):
    col = d[:, i]
    if np.all(np.isnan(col)): break
    print('%-28s median %7.0f  p90 %7.0f' % (labels[i], np.nanmedian(col), np.nanpercentile(col, 90)))
