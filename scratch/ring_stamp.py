"""s_memrealtime stamps of conv_ring workgroups (library built by scratch/build_variant.py stamp -DHR_RING_STAMP)."""
import os, sys, ctypes
os.environ['HRNET_HIP_LIB'] = '/root/repo/scratch/var_stamp/libhrnet_hip.so'
sys.path.insert(0, '/root/repo/tests'); sys.path.insert(0, '/root/repo/hrnet-hand-pose-estimation_amd/lib')
import torch, numpy as np
import hip_helpers as hh
from hipnet import _capi as C
dt = torch.bfloat16
L = C.lib()
L.hrnet_conv_ring_set_stamp.argtypes = [ctypes.c_void_p]

def case(N, H, W, Cin, Cout, mode):
    x = (torch.randn(N, H, W, Cin, device='cuda') * 1.5 + 0.3).to(dt)
    w = torch.randn(Cout, Cin, 3, 3) * 0.05
    wp, cop, cip = hh.pack_weights(w, dt)
    sums = torch.rand(8, 2, Cin, device='cuda') + 1.0
    sums[:, 1] += 10
    gb = torch.cat([torch.rand(Cin, device='cuda') + 0.5, torch.rand(Cin, device='cuda') - 0.5]).contiguous()
    y = torch.empty(N, H, W, cop, device='cuda', dtype=dt); st = torch.zeros(8, 2, cop, device='cuda')
    stamp = torch.zeros(2048 * 32, dtype=torch.int64, device='cuda')
    def run():
        if mode == 'raw':
            C.call('hrnet_conv2d_bnref', 1, x.data_ptr(), wp.data_ptr(), None, None, None, 0.0, 0.0, None, y.data_ptr(), st.data_ptr(),
                   N, H, W, Cin, H, W, cop, 3, 1, 0, C.stream_ptr())
        else:
            C.call('hrnet_conv2d_bnref', 1, x.data_ptr(), wp.data_ptr(), sums.data_ptr(), gb.data_ptr(), gb.data_ptr() + 4 * Cin, 1.0 / (N * H * W), 1e-5,
                   None, y.data_ptr(), st.data_ptr(), N, H, W, Cin, H, W, cop, 3, 1, 1, C.stream_ptr())
    for _ in range(5): run()
    torch.cuda.synchronize()
    L.hrnet_conv_ring_set_stamp(stamp.data_ptr())
    run(); torch.cuda.synchronize()
    L.hrnet_conv_ring_set_stamp(None)
    s = stamp.cpu().numpy().reshape(-1, 32)
    s = s[s[:, 0] > 0]
    t0 = s[:, 0].min()
    ns = (s - t0) * 10.0          # 100 MHz
    ns[s == 0] = np.nan
    print('%s N%d %dx%d %d->%d: %d workgroups; stamps (us, median over workgroups | min | max):' % (mode, N, H, W, Cin, Cout, len(s)))
    for k in range(32):
        col = ns[:, k]
        if np.all(np.isnan(col)): break
        print('  %2d  %7.2f | %7.2f | %7.2f' % (k, np.nanmedian(col) / 1e3, np.nanmin(col) / 1e3, np.nanmax(col) / 1e3))
    last = np.nanmax(ns, axis=1)
    print('  end: median %.2f max %.2f us' % (np.median(last) / 1e3, last.max() / 1e3))

case(64, 64, 64, 32, 32, 'raw')
case(64, 64, 64, 32, 32, 'bn')
case(64, 32, 32, 64, 64, 'raw')
case(64, 32, 32, 64, 64, 'bn')
