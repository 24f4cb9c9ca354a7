"""micro-timing of hrnet_head_mix / hrnet_upsample_bilinear_t at the w32 B=64 head shapes"""
import ctypes, os, sys
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/hrnet-hand-pose-estimation_amd/lib'); sys.path.insert(0, '/root/repo/tests')
import torch
from hipnet import _capi as C
DT = torch.bfloat16
N, H, W, C0, Cout = 64, 64, 64, 32, 480
dev = 'cuda:0'
def pp(ts):
    a = (ctypes.c_void_p * max(1, len(ts)))()
    for k, t in enumerate(ts): a[k] = t.data_ptr()
    return a
def ip(v): return (ctypes.c_int * max(1, len(v)))(*v)
x0 = torch.randn(N, H, W, C0, device=dev).to(DT)
w0 = torch.randn(Cout, C0, device=dev).to(DT)
bias = torch.randn(Cout, device=dev)
ts = [torch.randn(N, H >> j, W >> j, Cout, device=dev).to(DT) for j in (1, 2, 3)]
y = torch.empty(N, H, W, Cout, device=dev, dtype=DT)
sums = torch.zeros(8, 2, Cout, device=dev)
def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1000
for nup in (0, 1, 2, 3):
    for st in (0, 1):
        f = lambda: C.call('hrnet_head_mix', 1, x0.data_ptr(), w0.data_ptr(), bias.data_ptr(), y.data_ptr(),
                           sums.data_ptr() if st else None, 0, pp(ts[:nup]), ip([H >> j for j in (1, 2, 3)][:nup]),
                           ip([W >> j for j in (1, 2, 3)][:nup]), nup, N, H, W, C0, Cout, 0, C.stream_ptr())
        print('head_mix nup={} stats={}: {:.1f} us'.format(nup, st, timeit(f)))
G = torch.randn(N, H, W, Cout, device=dev).to(DT)
outs = [torch.empty(N, H >> j, W >> j, Cout, device=dev, dtype=DT) for j in (1, 2, 3)]
for sel in ([0], [1], [2], [0, 1, 2]):
    for streamed in (0, 1):
        o = [outs[k] for k in sel]
        f = lambda: C.call('hrnet_upsample_bilinear_t', 1, G.data_ptr(), pp(o), ip([t.shape[1] for t in o]),
                           ip([t.shape[2] for t in o]), len(o), N, H, W, Cout, 0, streamed, C.stream_ptr())
        print('upsample_t outs={} streamed={}: {:.1f} us'.format(sel, streamed, timeit(f)))
# references: a copy of the same bytes, the old concat
src = torch.empty_like(y)
print('copy 252 MB: {:.1f} us'.format(timeit(lambda: y.copy_(src))))
