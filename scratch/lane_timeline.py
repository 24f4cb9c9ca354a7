"""Timeline of the recorded forward / backward programs UNDER the multi-lane overlap, without a profiler: a timing
event behind every op on its lane (hrnet_program_run_streams_timed). Prints phase walls, per-lane busy time, and
(--dump) every op with its lane, end time and the time since the previous op of its lane ended."""
import ctypes, os, sys
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/hrnet-hand-pose-estimation_amd/lib')
import numpy as np, torch
import bench as B
from hipnet import _capi as C, engine, synth

batch = int(os.environ.get('BATCH', '64'))
model, cfg, sd = B.build_model('bf16', 'RHD_HRNet_w32_bf16_train.yaml' if os.path.exists('/root/repo/hrnet-hand-pose-estimation_amd/experiments/RHD/RHD_HRNet_w32_bf16_train.yaml') else 'RHD_HRNet_w32_max_hmloss_v1.yaml')
model = model.cuda().train()
b = synth.rhd_batch(batch, seed=1234)
x = torch.from_numpy(b['imgs']).cuda(); gt = torch.from_numpy(b['heatmaps']).cuda()
from core.loss import HeatmapLoss
crit = HeatmapLoss()
opt = torch.optim.Adam if False else None
net = model.hip()
for _ in range(3):
    hm, _ = model(x); loss = crit(hm, gt); loss.backward()
torch.cuda.synchronize()

TIMED = {}
orig_run = engine.Program.run
def timed_run(self, lo=0, hi=None, streams=None):
    if streams is None: return orig_run(self, lo, hi, streams)
    if self._arr is None: self.finalize()
    hi = len(self.ops) if hi is None else hi
    if hi <= lo: return
    base = ctypes.cast(ctypes.byref(self._arr, lo * ctypes.sizeof(C.HrOp)), ctypes.POINTER(C.HrOp))
    handles = (ctypes.c_void_p * len(streams))(*[C.stream_ptr() if st is None else st.cuda_stream for st in streams])
    out = (ctypes.c_float * (hi - lo))()
    torch.cuda.synchronize()
    C.call('hrnet_program_run_streams_timed', base, hi - lo, handles, len(streams), out)
    TIMED[id(self)] = (lo, np.array(out[:]))
engine.Program.run = timed_run
hm, _ = model(x); loss = crit(hm, gt); loss.backward()
torch.cuda.synchronize()
engine.Program.run = orig_run
plan = net.plan(batch, 256, 256, True, True)
KN = {1: 'conv', 2: 'wgrad', 5: 'sum', 6: 'grad_term', 7: 'bn_red', 8: 'bn_bfin', 9: 'cat', 10: 'cat_bwd', 11: 'im2col', 26: 'head_mix', 27: 'upsample_t', 28: 'head_bwd', 29: 'pool_red', 25: 'ew_table', 12: 'to_nchw', 13: 'to_nhwc',
      16: 'fill', 18: 'ev_rec', 19: 'ev_wait', 20: 'wred', 21: 'fused', 22: 'bn_fin_tab', 23: 'pw_fused', 24: 'conv_sum', 15: 'bias_grad'}
for pname, prog in (('fwd', plan.fwd), ('bwd', plan.bwd)):
    lo, t = TIMED[id(prog)]
    n = len(t)
    lanes = [int(prog.ops[lo + k].i[C.LANE_SLOT]) for k in range(n)]
    kinds = [int(prog.ops[lo + k].kind) for k in range(n)]
    prev = {}
    dur = np.zeros(n)
    for k in range(n):
        dur[k] = t[k] - prev.get(lanes[k], 0.0)       # time since the previous op of the lane ended (includes waits)
        prev[lanes[k]] = t[k]
    print('== %s: %d ops, wall %.3f ms' % (pname, n, t.max()))
    for l in sorted(set(lanes)):
        idx = [k for k in range(n) if lanes[k] == l and kinds[k] not in (18, 19)]
        print('   lane %d: %4d kernels, last end %.3f ms' % (l, len(idx), max(t[k] for k in idx) if idx else 0))
    # phases by tag prefix (lane 0 ops only carry the order)
    tag = None
    marks = []
    for k in range(n):
        tg = prog.tags.get(lo + k)
        if tg:
            ph = tg.split('.')[0] if not tg.startswith('stage') else '.'.join(tg.split('.')[:2])
            if ph != tag:
                marks.append((ph, k)); tag = ph
    seen = {}
    for ph, k in marks:
        seen.setdefault(ph, []).append(k)
    order = []
    for ph, k in marks:
        if ph not in order: order.append(ph)
    print('   phase first-op end times (ms):')
    for ph in order:
        ks = seen[ph]
        print('     %-14s first op %5d ends %7.3f   last tagged op %5d ends %7.3f' % (ph, ks[0], t[ks[0]], ks[-1], t[ks[-1]]))
    if '--dump' in sys.argv:
        for k in range(n):
            if kinds[k] in (18, 19): continue
            print('%5d L%d %-10s %-34s end %8.3f  d %7.1f us' % (lo + k, lanes[k], KN.get(kinds[k], str(kinds[k])), prog.tags.get(lo + k, ''), t[k], dur[k] * 1e3))
