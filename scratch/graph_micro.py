import torch, time
x = torch.zeros(1024, device='cuda')
big = torch.zeros(64 * 64 * 64 * 32, device='cuda', dtype=torch.bfloat16)   # 17 MB tensor: a ~10 us kernel
def chain(t, n):
    for _ in range(n):
        t.add_(1)
for name, t in (('tiny', x), ('17MB', big)):
    for n in (200,):
        chain(t, n); torch.cuda.synchronize()
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        t0 = time.perf_counter(); e0.record(); chain(t, n); e1.record(); th = time.perf_counter() - t0; torch.cuda.synchronize()
        print('%s stream: %d kernels, GPU %.1f us/kernel, host enqueue %.1f us/kernel' % (name, n, e0.elapsed_time(e1) * 1e3 / n, th * 1e6 / n))
        g = torch.cuda.CUDAGraph()
        s = torch.cuda.Stream()
        with torch.cuda.stream(s):
            with torch.cuda.graph(g, stream=s):
                chain(t, n)
        torch.cuda.synchronize()
        g.replay(); torch.cuda.synchronize()
        e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
        print('%s graph : %d kernels, GPU %.1f us/kernel' % (name, n, e0.elapsed_time(e1) * 1e3 / n))
