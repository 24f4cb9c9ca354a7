"""Plan builder + executor for the HRNet hot path on MI355X.

The model module (models/pose_hrnet.py) owns PyTorch parameters with the reference's
state_dict names; this file turns (parameters, input shape, mode) into two recorded programs
of C-ABI ops (forward, backward) over NHWC device buffers and runs each with ONE host call
(hrnet_program_run). PyTorch is only the allocator / stream / autograd anchor.

Dataflow conventions
  Act   a device tensor [N,H,W,C] in the compute dtype. A conv output followed by BatchNorm is
        stored RAW (pre-BN); its consumers apply scale/shift(+ReLU) while loading, so the
        normalised tensor never makes an HBM round trip (reference: 7 separate kernels per
        BasicBlock, lib/models/pose_hrnet.py:41-57).
  Val   (act, bn, relu): the logical value relu?(bn?(act)) a consumer reads.
  Grad  every Act has a gradient buffer. For a raw+BN act it first accumulates the gradient
        w.r.t. the post-activation value, then BatchNorm backward rewrites it in place into the
        gradient w.r.t. the raw conv output (two-stage deterministic reductions).

Scheduling
  Lanes  ops carry a lane (HIP stream); EVENT_RECORD / STREAM_WAIT ops order them. The branches of a
         HighResolutionModule run on their own lanes, and the fuse-layer convs stay on the lane of the
         branch they read; only the sums wait for all lanes (HRNET_LANES=0: one stream).
  Fused BatchNorm-backward sums  the input-gradient conv that writes the LAST contribution to an
         activation gradient gathers (sum dz, sum dz*y) in its epilogue (hrnet_conv2d_bwdstats), which
         replaces the separate reduction pass for ~80 % of the BatchNorms (HRNET_FUSE_BWDSTATS=0: off).
  Fused block backward  the two 3x3 convs of a BasicBlock whose widths fit (hrnet_bwd_fused_supported) run their
         whole backward - BatchNorm-backward apply, weight gradient, input gradient, residual add, ReLU mask, the
         next BatchNorm's sums - as ONE launch each (hrnet_conv3x3_bwd_fused) instead of grad_term + wgrad +
         conv_bs (HRNET_FUSED_BWD=0: off).
  Batched slab sums  every layer keeps its own weight-gradient slab region; one table-driven launch per
         lane segment sums them (HRNET_BATCH_WRED=0: per-layer launches on shared scratch).
  Bucket marks  op indices at which every gradient above a flat offset is final (module boundaries and
         every ~16 MB on lane 0): hipnet.optim.GradSync issues its all-reduces there.
"""
import ctypes
import os

import torch

from . import _capi as C


def _round_up(x, m):
    return (x + m - 1) // m * m


# Environment switches. PRODUCT switches are read whenever a plan is recorded (INTEGRATION.md lists them). Every other
# HRNET_* variable this file knows is a MEASUREMENT switch - it changes which ops a recorded program holds (fusion on /
# off, lanes, deferral sizes ...) and exists for A/B runs and for the tests that compare a fused path with its unfused
# form - and is honoured only together with HRNET_MEASURE=1; without it the defaults below are what runs, and a
# variable that is set but ignored is reported once.
PRODUCT_ENV = ('HRNET_DETERMINISTIC', 'HRNET_WGRAD_ATOMIC', 'HRNET_DP_PLAN', 'HRNET_LANES')
_ignored = set()


def _knob(name, default):
    if name in PRODUCT_ENV or os.environ.get('HRNET_MEASURE', '0') == '1':
        return os.environ.get(name, default)
    if name in os.environ and name not in _ignored:
        _ignored.add(name)
        import warnings
        warnings.warn('{}={} is a measurement switch and is ignored without HRNET_MEASURE=1'.format(
            name, os.environ[name]))
    return default


class Act(object):
    __slots__ = ('name', 'N', 'H', 'W', 'C', 't', 'g', 'ginit', 'bn', 'bn_done', 'nuse', 'bwd_rows', 'gmasked')

    def __init__(self, name, N, H, W, Cc):
        self.name, self.N, self.H, self.W, self.C = name, N, H, W, Cc
        self.t = None       # torch tensor (storage)
        self.g = None       # gradient tensor
        self.ginit = False  # gradient buffer written yet (while recording the backward)
        self.bn = None      # BNRec if this is a raw conv output followed by BatchNorm
        self.bn_done = False
        self.nuse = 0       # number of consumers of the logical value
        self.bwd_rows = None  # (rows tensor, nrows): BN-backward sums gathered by a dgrad epilogue
        self.gmasked = False  # the gradient buffer already carries the ReLU mask of this activation (fused backward)

    @property
    def pixels(self):
        return self.N * self.H * self.W


class Val(object):
    __slots__ = ('act', 'bn', 'relu')

    def __init__(self, act, bn=None, relu=False):
        self.act, self.bn, self.relu = act, bn, relu


class BNRec(object):
    def __init__(self, prefix, mod, dev):
        self.prefix = prefix
        self.mod = mod
        c = mod.num_features
        self.C = c
        f = dict(dtype=torch.float32, device=dev)
        self.scale = torch.empty(c, **f)
        self.shift = torch.empty(c, **f)
        self.mean = torch.empty(c, **f)
        self.invstd = torch.empty(c, **f)
        self.coef = torch.empty(3 * c, **f)
        self.sums = None        # [8][2][C] batch sums (consumer-side BatchNorm; a slice of the plan's arena)
        self.count = 0.0        # elements per channel the sums cover


class ConvRec(object):
    def __init__(self, prefix, mod, stem=False):
        self.prefix, self.mod, self.stem = prefix, mod, stem
        w = mod.weight
        self.Cout, self.Cin, self.ks = w.shape[0], w.shape[1], w.shape[2]
        self.Cout_pad = _round_up(self.Cout, 16)
        self.Cin_pad = 32 if stem else _round_up(self.Cin, 8)
        self.wf = None   # packed forward weights
        self.wd = None   # packed dgrad weights


class Program(object):
    """A recorded list of HrOp, run with ONE C call. Every op carries a lane (= stream index);
    EVENT_RECORD / STREAM_WAIT ops express the dependencies between lanes: the branches of a
    HighResolutionModule are independent chains, and weight-gradient work is off the backward
    critical path, so the latency-bound kernels of one lane fill the gaps of another. On a single
    stream (no side streams given) the same list runs in order and the event ops are skipped."""

    def __init__(self):
        self.ops = []
        self.lane = 0
        self.events = []
        self._arr = None
        self.tags = {}          # op index -> layer name (measurement scripts only)
        self.before_add = None  # called before every op is appended (the plan flushes a held-back sum there)

    def add(self, kind, ints=(), floats=(), ptrs=(), lane=None):
        if self.before_add is not None:
            self.before_add(self.lane if lane is None else lane)
        op = C.HrOp()
        op.kind = kind
        for k, v in enumerate(ints):
            op.i[k] = int(v)
        for k, v in enumerate(floats):
            op.f[k] = float(v)
        for k, v in enumerate(ptrs):
            op.p[k] = v
        op.i[C.LANE_SLOT] = self.lane if lane is None else lane
        self.ops.append(op)
        self._arr = None
        return len(self.ops) - 1

    def lane_of(self, idx):
        return self.ops[idx].i[C.LANE_SLOT]

    def sync(self, src, dst):
        """everything enqueued so far on lane `src` happens before what follows on lane `dst`"""
        ev = C.lib().hrnet_event_create()
        if not ev:
            raise RuntimeError('hrnet_event_create failed')
        self.events.append(ev)
        self.add(C.OP_EVENT_RECORD, ptrs=(ev,), lane=src)
        self.add(C.OP_STREAM_WAIT, ptrs=(ev,), lane=dst)

    def fork(self, lanes):
        for l in lanes:
            self.sync(0, l)

    def join(self, lanes):
        for l in lanes:
            self.sync(l, 0)

    def finalize(self):
        self._arr = (C.HrOp * len(self.ops))(*self.ops) if self.ops else None
        return self

    def set_ptr(self, op_index, slot, value):
        self._arr[op_index].p[slot] = value

    def set_int(self, op_index, slot, value):
        self._arr[op_index].i[slot] = value

    def run(self, lo=0, hi=None, streams=None):
        if self._arr is None:
            self.finalize()
        hi = len(self.ops) if hi is None else hi
        if hi <= lo:
            return
        base = ctypes.cast(ctypes.byref(self._arr, lo * ctypes.sizeof(C.HrOp)), ctypes.POINTER(C.HrOp))
        if streams is None:
            C.call('hrnet_program_run', base, hi - lo, C.stream_ptr())
        else:
            handles = (ctypes.c_void_p * len(streams))(
                *[C.stream_ptr() if st is None else st.cuda_stream for st in streams])
            C.call('hrnet_program_run_streams', base, hi - lo, handles, len(streams))

    def __del__(self):
        try:
            for ev in self.events:
                C.lib().hrnet_event_destroy(ev)
        except Exception:
            pass

    def __len__(self):
        return len(self.ops)


class PlanTicket(object):
    """held by the autograd node of one training forward: marks the plan busy until the backward has run or the
    graph is dropped (then the node, and with it this ticket, is garbage-collected)"""

    def __init__(self, plan):
        self.plan, self.gen = plan, plan.gen
        plan.busy = True

    def valid(self):
        return self.plan.gen == self.gen

    def release(self):
        if self.plan is not None and self.plan.gen == self.gen:
            self.plan.busy = False
        self.plan = None

    def __del__(self):
        try:
            self.release()
        except Exception:
            pass


class Plan(object):
    """Forward (+ backward) programs of one (batch, height, width, mode) instance."""

    def __init__(self, net, N, H, W, training, need_grad):
        self.net = net
        self.N, self.H, self.W = N, H, W
        self.training, self.need_grad = training, need_grad
        self.dev = net.device
        self.dt = net.compute_dtype
        self.dtid = C.dtype_id(self.dt)
        self.esize = 4 if self.dt == torch.float32 else 2
        self.fwd = Program()
        self.bwd = Program()
        self.pack_f = Program()   # master f32 OIHW -> packed forward weights
        self.pack_d = Program()   # -> packed dgrad weights
        self.keep = []            # tensors kept alive
        self.tape = []
        self.acts = []
        self.max_stats = 0
        self.max_slab = 0
        self.max_bwd_part = 0
        self.pending = []         # (program, op_index, slot, kind) scratch pointers to patch
        self.bucket_marks = []    # backward op indices after which a gradient bucket is complete
        self.tape_lanes = []
        self.nlanes = 4 if _knob('HRNET_LANES', '1') != '0' else 1   # module branches
        # weight-gradient lane (off by default): '2' = only the weight gradients of lane 0 (the high-resolution
        # branch, whose HBM-bound elementwise passes make it the critical chain) move to their own lane; '1' =
        # all of them. Measured on MI355X: '2' shortens the dependency-only critical path of the backward
        # program from 14.4 to 12.3 ms but the step stays at 24.4 ms - the step is throughput-bound, not
        # dependency-bound - and '1' is slower.
        wl = _knob('HRNET_WLANE', '0')
        self.wlane = 4 if (self.nlanes > 1 and wl in ('1', '2')) else 0
        self.defer_lanes = int(_knob('HRNET_DEFER_LANES', '3'))      # side lanes the deferred weight gradients use
        self.wlane_all = wl == '1'
        self.streams = None
        self.gen = 0              # bumped by every forward run: a backward must see the generation it recorded
        self.busy = False         # a training forward ran and its backward has not (PlanTicket)
        self._build()

    # ---- allocation helpers -----------------------------------------------------------
    def _act(self, name, N, H, W, Cc, grad=True):
        a = Act(name, N, H, W, Cc)
        a.t = torch.empty(N * H * W * Cc * self.esize, dtype=torch.uint8, device=self.dev)
        if self.need_grad and grad:
            a.g = torch.empty(N * H * W * Cc * self.esize, dtype=torch.uint8, device=self.dev)
        self.acts.append(a)
        return a

    def _f32(self, n, zero=False):
        t = (torch.zeros if zero else torch.empty)(n, dtype=torch.float32, device=self.dev)
        self.keep.append(t)
        return t

    def _scratch(self, prog, idx, slot, kind):
        self.pending.append((prog, idx, slot, kind))

    def _tape(self, entry):
        self.tape.append(entry)
        self.tape_lanes.append(self.fwd.lane)

    # ---- forward recording -------------------------------------------------------------
    def conv(self, xin, crec, stride=1, bnrec=None, relu=False, name=None):
        """conv (+bias) [+ BatchNorm statistics]; returns the Val consumers read."""
        x = xin.act
        ks = 1 if crec.stem else crec.ks
        Ho = (x.H + 2 * (ks // 2) - ks) // stride + 1
        Wo = (x.W + 2 * (ks // 2) - ks) // stride + 1
        y = self._act(name or crec.prefix, x.N, Ho, Wo, crec.Cout_pad)
        cin = x.C
        assert cin == crec.Cin_pad, (crec.prefix, cin, crec.Cin_pad)
        want_stats = bnrec is not None and self.training
        tiles = C.call('hrnet_conv_tiles', x.N, Ho, Wo, crec.Cout_pad, ks, stride)
        bias = crec.mod.bias
        sums_in = self.bn_sums and xin.bn is not None       # the input's BatchNorm from its batch sums, on the fly
        m_in = xin.bn.mod if xin.bn is not None else None
        ps = self._pending_sum.get(self.fwd.lane)
        # (the wide layers' fat LDS-ring launches take one input tensor: their residual sums stay separate launches)
        fat = (ks == 3 and stride == 1 and cin >= 96 and (self.bn_sums or not want_stats)
               and C.call('hrnet_conv_ring_supported', self.dtid, x.N, x.H, x.W, cin, crec.Cout_pad) >= 3)
        if (ps is not None and ps['out'] is x and xin.bn is None and not xin.relu and stride == 1 and not crec.stem
                and bias is None and ps['lane'] == self.fwd.lane and not fat):
            # the residual sum that produced x has not been emitted: this conv forms it in its prologue and writes
            # it out on the side (hrnet_conv2d_sum) - one launch and one tensor read less per block
            del self._pending_sum[self.fwd.lane]
            bt, it = ps['bn_term'], ps['id_term']
            sm = self.bn_sums
            mb = bt.bn.mod
            i = self.fwd.add(C.OP_CONV_SUM,
                             ints=(self.dtid, x.N, x.H, x.W, cin, crec.Cout_pad, ks, 1 if (want_stats and sm) else 0),
                             floats=((1.0 / bt.bn.count, mb.eps) if sm else ()),
                             ptrs=(C.ptr(bt.act.t), C.ptr(crec.wf),
                                   None if sm else C.ptr(bt.bn.scale), None if sm else C.ptr(bt.bn.shift),
                                   C.ptr(bt.bn.sums) if sm else None, C.ptr(mb.weight) if sm else None,
                                   C.ptr(mb.bias) if sm else None,
                                   C.ptr(y.t), C.ptr(bnrec.sums) if (want_stats and sm) else None,
                                   C.ptr(it.act.t), C.ptr(x.t)))
            self.n_fused_sums += 1
            stats_slot = 8
        else:
            stats_slot = 6
            i = self.fwd.add(C.OP_CONV,
                             ints=(self.dtid, x.N, x.H, x.W, cin, Ho, Wo, crec.Cout_pad, ks, stride, 0,
                                   1 if xin.relu else 0, 0, 1 if (want_stats and self.bn_sums) else 0),
                             floats=((1.0 / xin.bn.count, m_in.eps) if sums_in else ()),
                             ptrs=(C.ptr(x.t), C.ptr(crec.wf),
                                   C.ptr(xin.bn.scale) if (xin.bn and not sums_in) else None,
                                   C.ptr(xin.bn.shift) if (xin.bn and not sums_in) else None,
                                   C.ptr(self.net.bias_pad[crec.prefix]) if bias is not None else None,
                                   C.ptr(y.t), C.ptr(bnrec.sums) if (want_stats and self.bn_sums) else None,
                                   None, None, None, None,
                                   C.ptr(xin.bn.sums) if sums_in else None, C.ptr(m_in.weight) if sums_in else None,
                                   C.ptr(m_in.bias) if sums_in else None))
        self.fwd.tags[i] = crec.prefix
        if want_stats and not self.bn_sums:
            self.max_stats = max(self.max_stats, tiles * 2 * crec.Cout_pad)
            self._scratch(self.fwd, i, stats_slot, 'stats')
        x.nuse += 1
        if bnrec is not None:
            y.bn = bnrec
            bnrec.count = float(y.pixels)
            if self.training and self.bn_sums:
                self.bn_finalize_list.append(bnrec)      # one table-driven launch at the end of the pass
            elif self.training:
                m = bnrec.mod
                j = self.fwd.add(C.OP_BN_FINALIZE, ints=(tiles, bnrec.C, 1),
                                 floats=(y.pixels, m.momentum if m.momentum is not None else 0.1, m.eps),
                                 ptrs=(None, C.ptr(m.weight), C.ptr(m.bias), C.ptr(m.running_mean),
                                       C.ptr(m.running_var), C.ptr(m.num_batches_tracked), C.ptr(bnrec.scale),
                                       C.ptr(bnrec.shift), C.ptr(bnrec.mean), C.ptr(bnrec.invstd)))
                self._scratch(self.fwd, j, 0, 'stats')
        self._tape(('conv', xin, crec, stride, y, bnrec))
        return Val(y, bnrec, relu)

    def sum(self, terms, shifts, relu_out, name, batch=None):
        """batch: a list - the op is appended to it as a job of a table-driven launch (HR_OP_EW_TABLE) instead of
        being emitted as a launch of its own"""
        t0 = terms[0].act
        sh0 = shifts[0]
        out = self._act(name, t0.N, t0.H << sh0, t0.W << sh0, t0.C)
        ints = [self.dtid, out.N, out.H, out.W, out.C, len(terms), 1 if relu_out else 0]
        ints += list(shifts) + [0] * (4 - len(terms))
        ints += [1 if t.relu else 0 for t in terms] + [0] * (4 - len(terms))
        ptrs = [C.ptr(out.t)]
        ptrs += [C.ptr(t.act.t) for t in terms] + [None] * (4 - len(terms))
        # relu(bn(y) + identity) that closes a block: held back - if the very next op of this lane is the stride-1
        # conv that reads it, that conv forms the sum itself (hrnet_conv2d_sum); any other op emits it first
        bn_terms = [t for t in terms if t.bn is not None]
        hold = (self.fuse_sums and len(terms) == 2 and list(shifts) == [0, 0] and relu_out and len(bn_terms) == 1
                and not bn_terms[0].relu and all(t.bn is not None or not t.relu for t in terms)
                and (not self.bn_sums or t0.C <= 768) and batch is None)
        emit = self.fwd.add
        if hold:
            held = []
            emit = lambda *a, **k: held.append((a, k))
        elif batch is not None:
            def emit(kind, ints=(), floats=(), ptrs=()):
                op = C.HrOp()
                op.kind = kind
                for k, v in enumerate(ints):
                    op.i[k] = int(v)
                for k, v in enumerate(floats):
                    op.f[k] = float(v)
                for k, v in enumerate(ptrs):
                    op.p[k] = v
                batch.append(op)
        if self.bn_sums and any(t.bn for t in terms):
            import struct
            mode = sum(1 << k for k, t in enumerate(terms) if t.bn)
            ints += [mode, struct.unpack('i', struct.pack('f', float(next(t.bn.mod.eps for t in terms if t.bn))))[0]]
            ptrs += [C.ptr(t.bn.sums) if t.bn else None for t in terms] + [None] * (4 - len(terms))
            ptrs += [C.ptr(t.bn.mod.weight) if t.bn else None for t in terms] + [None] * (4 - len(terms))
            floats = [1.0 / t.bn.count if t.bn else 0.0 for t in terms] + [0.0] * (4 - len(terms))
            emit(C.OP_SUM_TERMS, ints=ints, floats=floats, ptrs=ptrs)
        else:
            ptrs += [C.ptr(t.bn.scale) if t.bn else None for t in terms] + [None] * (4 - len(terms))
            ptrs += [C.ptr(t.bn.shift) if t.bn else None for t in terms] + [None] * (4 - len(terms))
            emit(C.OP_SUM_TERMS, ints=ints, ptrs=ptrs)
        if hold:
            self._flush_pending_sum(self.fwd.lane)
            self._pending_sum[self.fwd.lane] = dict(out=out, lane=self.fwd.lane, op=held[0], bn_term=bn_terms[0],
                                                    id_term=next(t for t in terms if t.bn is None))
        for t in terms:
            t.act.nuse += 1
        self._tape(('sum', list(terms), list(shifts), relu_out, out))
        return Val(out)

    def _flush_pending_sum(self, lane=None):
        """emit the held-back residual sum of `lane` (None: of every lane) as its own launch (its consumer was not a
        fusable conv). One slot per lane: the lanes of a module may be recorded interleaved."""
        for l in ([lane] if lane is not None else sorted(self._pending_sum)):
            ps = self._pending_sum.pop(l, None)
            if ps is not None:
                a, k = ps['op']
                keep = self.fwd.lane
                self.fwd.lane = ps['lane']
                self.fwd.add(*a, **k)
                self.fwd.lane = keep

    def bilinear_cat(self, vals, name, align=False):
        a0 = vals[0].act
        ctot = sum(v.act.C for v in vals)
        cat = self._act(name, a0.N, a0.H, a0.W, ctot)
        hs = [v.act.H for v in vals] + [0] * (4 - len(vals))
        ws = [v.act.W for v in vals] + [0] * (4 - len(vals))
        cs = [v.act.C for v in vals] + [0] * (4 - len(vals))
        self.fwd.add(C.OP_BILINEAR_CAT, ints=[self.dtid, len(vals), a0.N, a0.H, a0.W] + hs + ws + cs,
                     floats=(1.0 if align else 0.0,), ptrs=[C.ptr(cat.t)] + [C.ptr(v.act.t) for v in vals])
        for v in vals:
            assert v.bn is None and not v.relu
            v.act.nuse += 1
        self._tape(('cat', list(vals), cat, align))
        return Val(cat)

    def head_mix(self, vals, crec, bnrec, align=False):
        """last_layer[0] over cat(x0, up(x1), ...) WITHOUT the concat (pose_hrnet.py:560-566; csrc/gemm_pw.hip):
        t_j = W_j x_j at branch j's resolution, then y = W0 x0 + bias + sum_j up(t_j) with its batch statistics."""
        net = self.net
        x0 = vals[0].act
        widths = [v.act.C for v in vals]
        offs = [0]
        for c in widths:
            offs.append(offs[-1] + c)
        wfs = net.head_slices(crec, widths)
        y = self._act(crec.prefix, x0.N, x0.H, x0.W, crec.Cout_pad)
        ts = []
        for j in range(1, len(vals)):
            xa = vals[j].act
            t = self._act('{}.t{}'.format(crec.prefix, j), xa.N, xa.H, xa.W, crec.Cout_pad)
            i = self.fwd.add(C.OP_CONV, ints=(self.dtid, xa.N, xa.H, xa.W, xa.C, xa.H, xa.W, crec.Cout_pad, 1, 1, 0,
                                              0, 0, 0),
                             ptrs=(C.ptr(xa.t), C.ptr(wfs[j]), None, None, None, C.ptr(t.t), None))
            self.fwd.tags[i] = '{}.t{}'.format(crec.prefix, j)
            ts.append(t)
        want_stats = bnrec is not None and self.training
        rows_mode = 0 if self.bn_sums else 1
        ints = [self.dtid, x0.N, x0.H, x0.W, x0.C, crec.Cout_pad, len(ts), 1 if align else 0] + [0] * 6 + [rows_mode]
        for k, t in enumerate(ts):
            ints[8 + 2 * k], ints[9 + 2 * k] = t.H, t.W
        bias = crec.mod.bias
        i = self.fwd.add(C.OP_HEAD_MIX, ints=ints,
                         ptrs=[C.ptr(x0.t), C.ptr(wfs[0]), C.ptr(net.bias_pad[crec.prefix]) if bias is not None else None,
                               C.ptr(y.t), C.ptr(bnrec.sums) if (want_stats and self.bn_sums) else None]
                         + [C.ptr(t.t) for t in ts])
        self.fwd.tags[i] = crec.prefix
        rows = C.call('hrnet_head_mix_rows', x0.N, x0.H, x0.W)
        if want_stats and not self.bn_sums:
            self.max_stats = max(self.max_stats, rows * 2 * crec.Cout_pad)
            self._scratch(self.fwd, i, 4, 'stats')
        for v in vals:
            assert v.bn is None and not v.relu
            v.act.nuse += 1
        if bnrec is not None:
            y.bn = bnrec
            bnrec.count = float(y.pixels)
            if self.training and self.bn_sums:
                self.bn_finalize_list.append(bnrec)
            elif self.training:
                m = bnrec.mod
                j = self.fwd.add(C.OP_BN_FINALIZE, ints=(rows, bnrec.C, 1),
                                 floats=(y.pixels, m.momentum if m.momentum is not None else 0.1, m.eps),
                                 ptrs=(None, C.ptr(m.weight), C.ptr(m.bias), C.ptr(m.running_mean),
                                       C.ptr(m.running_var), C.ptr(m.num_batches_tracked), C.ptr(bnrec.scale),
                                       C.ptr(bnrec.shift), C.ptr(bnrec.mean), C.ptr(bnrec.invstd)))
                self._scratch(self.fwd, j, 0, 'stats')
        self._tape(('headmix', list(vals), crec, y, bnrec, ts, offs, align))
        self.n_head_mix = 1
        self._head_y = y
        return Val(y, bnrec, True)

    # ---- network walk (reference: PoseHighResolutionNet.forward, pose_hrnet.py:511-568) ----
    def _build(self):
        net = self.net
        N, H, W = self.N, self.H, self.W
        # BatchNorm coefficient buffers (scale/shift/mean/invstd/coef) belong to the PLAN: two plans of one net
        # can be in flight at once (hipnet.net.HipNet.plan), and each backward needs its own forward's statistics
        self.bns = {name: BNRec(name, rec.mod, self.dev) for name, rec in net.bns.items()}
        # residual sums fused into the conv that reads them (hrnet_conv2d_sum): HRNET_FUSE_SUM=0 turns it off.
        # Training only: 18.75 vs 18.92 ms/step; an eval pass is faster with the separate sum kernels (5.25 vs
        # 5.36 ms: the two-tensor prologue re-reads two halos and lengthens the latency-bound conv launches)
        self.fuse_sums = self.training and _knob('HRNET_FUSE_SUM', '1') != '0'
        self._pending_sum = {}        # lane -> held-back residual sum
        self.n_fused_sums = 0
        self.n_batched_fwd_sums = 0
        self.fwd.before_add = self._flush_pending_sum
        # Consumer-side BatchNorm (training): producers add their batch sums into 8 partial copies per BatchNorm with
        # float atomics, forward consumers build scale/shift from them on the fly, and ONE table-driven launch at the
        # end of the pass fills the arrays the backward pass reads and updates the running statistics - instead of a
        # finalize launch after each of the 306 convs. HRNET_DETERMINISTIC=1: per-workgroup rows + a finalize launch
        # per BatchNorm (bit-reproducible statistics).
        contiguous = all(b.mod.bias.data_ptr() == b.mod.weight.data_ptr() + 4 * b.C for b in self.bns.values())
        # capacity of the on-the-fly coefficient tables: 768 channels in the conv prologues (HR_CONV_MAXC: the head's
        # 480 / 720), 384 in hrnet_sum_terms (SUM_MAXC: the widest BatchNorm that reaches a sum is a branch width -
        # the head's BatchNorm is read by a conv); wider nets fall back to per-BatchNorm finalize launches
        sum_w = max([b.C for n, b in self.bns.items() if not n.startswith('last_layer')] or [0])
        self.bn_sums = (self.training and contiguous and _knob('HRNET_DETERMINISTIC', '0') != '1'
                        and max(b.C for b in self.bns.values()) <= 768 and sum_w <= 384)
        self.bn_finalize_list = []
        if self.bn_sums:
            total = sum(16 * b.C for b in self.bns.values())
            self.bn_arena = self._f32(total, zero=True)
            off = 0
            for b in self.bns.values():
                b.sums = self.bn_arena[off:off + 16 * b.C]
                off += 16 * b.C
            nbytes = total * 4
            self.fwd.add(C.OP_FILL, ints=(nbytes & 0xffffffff, nbytes >> 32), ptrs=(C.ptr(self.bn_arena),))
        cv, bn = net.convs, self.bns
        if not self.training:
            # eval mode: every BatchNorm's affine from its running statistics, ONE table launch (306 launches of
            # 12 us each before: a third of the isolated kernel time of an eval forward pass)
            self.bn_finalize_list = list(self.bns.values())
            self._add_bn_finalize_table(eval_mode=True)
            self.bn_finalize_list = []
        Ho, Wo = (H + 1) // 2, (W + 1) // 2
        cols = self._act('stem.cols', N, Ho, Wo, 32, grad=False)
        self.in_op = self.fwd.add(C.OP_IM2COL_STEM, ints=(self.dtid, N, 3, H, W, Ho, Wo, 32),
                                  ptrs=(None, C.ptr(cols.t)))
        v = self.conv(Val(cols), cv['conv1'], 1, bn['bn1'], relu=True)
        v = self.conv(v, cv['conv2'], 2, bn['bn2'], relu=True)
        for k in range(4):
            p = 'layer1.{}'.format(k)
            a = self.conv(v, cv[p + '.conv1'], 1, bn[p + '.bn1'], relu=True)
            b = self.conv(a, cv[p + '.conv2'], 1, bn[p + '.bn2'], relu=True)
            c = self.conv(b, cv[p + '.conv3'], 1, bn[p + '.bn3'], relu=False)
            if (p + '.downsample.0') in cv:
                res = self.conv(v, cv[p + '.downsample.0'], 1, bn[p + '.downsample.1'], relu=False)
            else:
                res = v
            v = self.sum([c, res], [0, 0], True, p + '.out')
        ys = [v]
        inter = None
        for s in (2, 3, 4):
            sc = net.stage_cfg[s]
            nbr = sc['NUM_BRANCHES']
            tp = 'transition{}'.format(s - 1)
            xs = []
            for i in range(nbr):
                if i < len(ys):
                    key = '{}.{}.0'.format(tp, i)
                    xs.append(self.conv(ys[i], cv[key], 1, bn['{}.{}.1'.format(tp, i)], relu=True)
                              if key in cv else ys[i])
                else:
                    t = ys[-1]
                    for j in range(i + 1 - len(ys)):
                        q = '{}.{}.{}'.format(tp, i, j)
                        t = self.conv(t, cv[q + '.0'], 2, bn[q + '.1'], relu=True)
                    xs.append(t)
            for m in range(sc['NUM_MODULES']):
                xs = self._hr_module(xs, 'stage{}.{}'.format(s, m), sc['NUM_BLOCKS'])
            ys = xs
            if s == 3:
                inter = ys[0]
        # pose_hrnet_softmax (lib/models/pose_hrnet_softmax.py:499-506): align_corners=True, inter_feat = the concat
        # The head: last_layer[0] commutes with the bilinear upsampling, so (bf16, widths the mix launch takes) the
        # 480-channel concat and its gradient are never formed - head_mix(). HRNET_HEAD_MIX=0, the fp32 device path
        # and pose_hrnet_softmax (whose inter_feat IS the concat) keep the concat form.
        align = bool(getattr(net.module, 'head_align_corners', False))
        c0 = cv['last_layer.0']
        self.n_head_mix = 0
        self._head_y = None
        self._head_bwd = None
        mix = (_knob('HRNET_HEAD_MIX', '1') != '0' and not getattr(net.module, 'inter_from_cat', False)
               and 2 <= len(ys) <= 4 and all(v.bn is None and not v.relu for v in ys)
               and sum(v.act.C for v in ys) == c0.Cin and c0.Cin_pad == c0.Cin and c0.Cout_pad == c0.Cout
               and C.call('hrnet_head_mix_supported', self.dtid, ys[0].act.C, c0.Cout_pad) == 1)
        if mix:
            h = self.head_mix(ys, c0, bn['last_layer.1'], align=align)
        else:
            cat = self.bilinear_cat(ys, 'head.cat', align=align)
            if getattr(net.module, 'inter_from_cat', False):
                inter = cat
            h = self.conv(cat, c0, 1, bn['last_layer.1'], relu=True)
        out = self.conv(h, cv['last_layer.3'], 1, None, relu=False)
        self.out_act, self.inter_act = out.act, inter.act
        self.nj = cv['last_layer.3'].Cout
        self.out_op = self.fwd.add(C.OP_NHWC_TO_NCHW, ints=(self.dtid, N, out.act.H, out.act.W, out.act.C, self.nj),
                                   ptrs=(C.ptr(out.act.t), None))
        self.inter_op = self.fwd.add(C.OP_NHWC_TO_NCHW,
                                     ints=(self.dtid, N, inter.act.H, inter.act.W, inter.act.C, inter.act.C),
                                     ptrs=(C.ptr(inter.act.t), None))
        if self.bn_sums and self.bn_finalize_list:
            self._add_bn_finalize_table()
        self._flush_pending_sum()
        self.fwd.before_add = None
        if self.need_grad:
            self._build_backward()
            if self.wlane:
                self.bwd.sync(self.wlane, 0)     # all weight gradients done before the optimiser
        self._resolve_scratch()
        self.fwd.finalize()
        self.bwd.finalize()

    def _add_bn_finalize_table(self, eval_mode=False):
        ents = (C.HrBnEnt * len(self.bn_finalize_list))()
        block = 0
        for e, b in zip(ents, self.bn_finalize_list):
            m = b.mod
            e.sums, e.gamma, e.beta = (None if eval_mode else C.ptr(b.sums)), C.ptr(m.weight), C.ptr(m.bias)
            e.running_mean, e.running_var = C.ptr(m.running_mean), C.ptr(m.running_var)
            e.num_batches_tracked = None if eval_mode else C.ptr(m.num_batches_tracked)
            e.scale, e.shift, e.mean, e.invstd = C.ptr(b.scale), C.ptr(b.shift), C.ptr(b.mean), C.ptr(b.invstd)
            e.count = 1.0 if eval_mode else b.count
            e.momentum, e.eps = (m.momentum if m.momentum is not None else 0.1), m.eps
            e.C, e.block0 = b.C, block
            block += (b.C + 255) // 256
        raw = bytes(ctypes.string_at(ctypes.addressof(ents), ctypes.sizeof(ents)))
        table = torch.frombuffer(bytearray(raw), dtype=torch.uint8).to(self.dev)
        self.keep.append(table)
        self.fwd.lane = 0
        self.fwd.add(C.OP_BN_FINALIZE_TABLE, ints=(len(self.bn_finalize_list), block), ptrs=(C.ptr(table),))

    def _hr_module(self, xs, pre, num_blocks):
        """HighResolutionModule.forward, pose_hrnet.py:247-266."""
        cv, bn = self.net.convs, self.bns
        nb = len(xs)
        xs = list(xs)
        side = [i for i in range(1, nb) if i < self.nlanes]
        if side:
            self.fwd.fork(side)
            self._tape(('fork', side))
        # recording order of the branches: lane by lane (every block of branch 0, then branch 1 ...), or - HRNET_INTERLEAVE=1,
        # measurement - block by block across the branches, so that the host hands every lane its first launches at once
        # (the programs are enqueued in recording order at ~3.5 us per op)
        interleave = _knob('HRNET_INTERLEAVE', '0') == '1' and len(side) > 0
        order = ([(i, k) for k in range(max(num_blocks[:nb])) for i in range(nb) if k < num_blocks[i]] if interleave
                 else [(i, k) for i in range(nb) for k in range(num_blocks[i])])
        for i, k in order:
            self.fwd.lane = i if i in side else 0
            b = '{}.branches.{}.{}'.format(pre, i, k)
            a = self.conv(xs[i], cv[b + '.conv1'], 1, bn[b + '.bn1'], relu=True)
            c = self.conv(a, cv[b + '.conv2'], 1, bn[b + '.bn2'], relu=False)
            xs[i] = self.sum([c, xs[i]], [0, 0], True, b + '.out')
        # fuse-layer convolutions stay on the lane of the branch they READ (source-major): they start as
        # soon as that branch is done, and in the backward pass every accumulation into a branch
        # output's gradient is ordered on one lane. The sums wait for all lanes.
        fuse_lanes = _knob('HRNET_FUSE_LANES', '1') != '0'
        if side and not fuse_lanes:
            self.fwd.lane = 0
            self.fwd.join(side)
            self._tape(('join', side))
        term_of = {}
        for j in range(nb):
            if fuse_lanes:
                self.fwd.lane = j if j in side else 0
            for i in range(nb):
                if j == i:
                    term_of[(i, j)] = (xs[j], 0)
                elif j > i:
                    f = '{}.fuse_layers.{}.{}'.format(pre, i, j)
                    term_of[(i, j)] = (self.conv(xs[j], cv[f + '.0'], 1, bn[f + '.1'], relu=False), j - i)
                else:
                    t = xs[j]
                    for k in range(i - j):
                        f = '{}.fuse_layers.{}.{}.{}'.format(pre, i, j, k)
                        t = self.conv(t, cv[f + '.0'], 2, bn[f + '.1'], relu=(k != i - j - 1))
                    term_of[(i, j)] = (t, 0)
        self.fwd.lane = 0
        if side and fuse_lanes:
            self.fwd.join(side)
            self._tape(('join', side))
        # the sums of the module outputs run on the lane of their output branch (a second fork/join): in the
        # backward pass that spreads a module's 16 sum-term passes over the lanes, and every consumer of a
        # branch gradient sits on that branch's lane
        sum_lanes = side and fuse_lanes and _knob('HRNET_SUM_LANES', '1') != '0'
        # ... or, forward (HRNET_BATCH_SUMFWD, default on): the nb sums as ONE table-driven launch on lane 0 - they are
        # 10-30 us each, and the fork / join around them cost more than running them side by side saved. The tape
        # keeps the ('fork' / 'join', side, 'sums') markers: the backward pass batches its side of them the same way.
        batch = [] if (sum_lanes and _knob('HRNET_BATCH_SUMFWD', '1') != '0') else None
        if sum_lanes:
            if batch is None:
                self.fwd.fork(side)
            self._tape(('fork', side, 'sums'))
        outs = []
        for i in range(nb):
            if sum_lanes and batch is None:
                self.fwd.lane = i if i in side else 0
            terms = [term_of[(i, j)][0] for j in range(nb)]
            shifts = [term_of[(i, j)][1] for j in range(nb)]
            # the output-resolution term first (sum_terms sizes the output from term 0)
            order = sorted(range(nb), key=lambda q: shifts[q])
            outs.append(self.sum([terms[q] for q in order], [shifts[q] for q in order], True,
                                 '{}.fuse.{}'.format(pre, i), batch=batch))
            if batch is not None:
                self.tape_lanes[-1] = i if i in side else 0      # (the lane the backward pass works this branch on)
        self.fwd.lane = 0
        if batch:
            block, sums_mode = 0, 0
            for op in batch:
                nb_ = C.call('hrnet_ew_table_blocks', C.OP_SUM_TERMS, self.dtid, op.i[1], op.i[2], op.i[3], op.i[4])
                # (slots of a table job: i[16] = first block, i[17] = blocks; the eps of HR_OP_SUM_TERMS moves to i[18])
                op.i[18] = op.i[16]
                op.i[16], op.i[17] = block, nb_
                block += nb_
                sums_mode |= op.i[15]
            arr = (C.HrOp * len(batch))(*batch)
            raw = bytes(ctypes.string_at(ctypes.addressof(arr), ctypes.sizeof(arr)))
            table = torch.frombuffer(bytearray(raw), dtype=torch.uint8).to(self.dev)
            self.keep.append(table)
            self.fwd.add(C.OP_EW_TABLE, ints=(len(batch), block, C.OP_SUM_TERMS, self.dtid, 1 if sums_mode else 0),
                         ptrs=(C.ptr(table),))
            self.n_batched_fwd_sums += len(batch)
        if sum_lanes:
            if batch is None:
                self.fwd.join(side)
            self._tape(('join', side, 'sums'))
        return outs

    # ---- backward recording ---------------------------------------------------------------
    def _bn_backward(self, y, g_src, mask, sh, inner_relu, extra=None, pooled=None):
        """BatchNorm backward for raw act y: reads the upstream gradient from g_src (pooled over
        2^sh blocks, masked by mask>0 and the BN's own ReLU), writes d(raw) into y.g. `extra`: an
        identity term of the same sum whose gradient (the same dz) is written by the same pass."""
        b = y.bn
        blocks = C.call('hrnet_reduce_blocks', y.N, y.H, y.W, y.C)
        self.max_bwd_part = max(self.max_bwd_part, blocks * 2 * y.C)
        m = b.mod
        # a reduction pass (no sums gathered by a producer) pools and masks the upstream gradient: it keeps that dz in
        # y.g, and the apply pass reads it from there instead of pooling and masking the full-resolution tensors
        # again (an up-sampled fuse term of branch 0 re-read 2 x 16.8 MB per term). HRNET_KEEP_DZ=0: both passes pool.
        keep_dz = None
        if y.bwd_rows is None and g_src != C.ptr(y.g) and _knob('HRNET_KEEP_DZ', '1') != '0':
            keep_dz = C.ptr(y.g)
        if pooled is not None:
            # the reduction ran as one level of a HR_OP_POOL_REDUCE job (its dz is in y.g, its partial rows in `pooled`)
            part, pblocks = pooled
            keep_dz = C.ptr(y.g)
            self._emit(C.OP_BN_BWD_FINALIZE, ints=(pblocks, y.C, 1), floats=(y.pixels,),
                       ptrs=(C.ptr(part), C.ptr(m.weight), C.ptr(b.mean), C.ptr(b.invstd),
                             C.ptr(self.net.grad_of(m.weight)), C.ptr(self.net.grad_of(m.bias)), C.ptr(b.coef)))
        if pooled is not None:
            pass
        elif y.bwd_rows is not None:
            # the dgrad conv that finished g_src already gathered (sum dz, sum dz*y) in its epilogue
            rows, blocks = y.bwd_rows
            assert sh == 0
            self._emit(C.OP_BN_BWD_FINALIZE, ints=(blocks, y.C, 1), floats=(y.pixels,),
                       ptrs=(C.ptr(rows), C.ptr(m.weight), C.ptr(b.mean), C.ptr(b.invstd),
                             C.ptr(self.net.grad_of(m.weight)), C.ptr(self.net.grad_of(m.bias)), C.ptr(b.coef)))
        elif self._batch is not None:
            part = self._f32(blocks * 2 * y.C)          # (a job of a batched launch: partial rows of its own)
            self._emit(C.OP_BN_BWD_REDUCE, ints=(self.dtid, y.N, y.H, y.W, y.C, sh, 1 if inner_relu else 0),
                       ptrs=(C.ptr(part), g_src, mask, C.ptr(y.t), C.ptr(b.scale), C.ptr(b.shift), keep_dz))
            self._emit(C.OP_BN_BWD_FINALIZE, ints=(blocks, y.C, 1), floats=(y.pixels,),
                       ptrs=(C.ptr(part), C.ptr(m.weight), C.ptr(b.mean), C.ptr(b.invstd),
                             C.ptr(self.net.grad_of(m.weight)), C.ptr(self.net.grad_of(m.bias)), C.ptr(b.coef)))
        else:
            i = self.bwd.add(C.OP_BN_BWD_REDUCE, ints=(self.dtid, y.N, y.H, y.W, y.C, sh, 1 if inner_relu else 0),
                             ptrs=(None, g_src, mask, C.ptr(y.t), C.ptr(b.scale), C.ptr(b.shift), keep_dz))
            self._scratch(self.bwd, i, 0, 'bwdpart')
            j = self.bwd.add(C.OP_BN_BWD_FINALIZE, ints=(blocks, y.C, 1), floats=(y.pixels,),
                             ptrs=(None, C.ptr(m.weight), C.ptr(b.mean), C.ptr(b.invstd),
                                   C.ptr(self.net.grad_of(m.weight)), C.ptr(self.net.grad_of(m.bias)), C.ptr(b.coef)))
            self._scratch(self.bwd, j, 0, 'bwdpart')
        ints = [self.dtid, y.N, y.H, y.W, y.C, sh, 1 if inner_relu else 0, 0, 0]
        ptrs = [C.ptr(y.g), g_src, mask, C.ptr(y.t), C.ptr(b.scale), C.ptr(b.shift), C.ptr(b.coef), None]
        if keep_dz is not None:
            ints[5], ints[6] = 0, 0                   # dz as the reduction stored it: pooled and masked already
            ptrs[1], ptrs[2] = keep_dz, None
        if extra is not None:
            assert sh == 0 and not inner_relu
            ints[8] = 1 if extra.ginit else 0
            ptrs[7] = C.ptr(extra.g)
            extra.ginit = True
        self._emit(C.OP_GRAD_TERM, ints=ints, ptrs=ptrs)
        y.bn_done = True
        y.ginit = True

    # ---- batched element-wise jobs (the backward of a HighResolutionModule's fuse sums) ----
    def _emit(self, kind, ints=(), floats=(), ptrs=()):
        """an HR_OP_GRAD_TERM / HR_OP_BN_BWD_REDUCE / HR_OP_BN_BWD_FINALIZE op: its own launch, or - while a batch is
        open - a job of the batched launch of its kind (HR_OP_EW_TABLE)"""
        if self._batch is None:
            return self.bwd.add(kind, ints=ints, floats=floats, ptrs=ptrs)
        op = C.HrOp()
        op.kind = kind
        for k, v in enumerate(ints):
            op.i[k] = int(v)
        for k, v in enumerate(floats):
            op.f[k] = float(v)
        for k, v in enumerate(ptrs):
            op.p[k] = v
        self._batch[kind].append(op)
        return None

    def _flush_batch(self):
        """the open batch as (at most) three launches on lane 0: every reduction, every finalize, every apply job"""
        batch, self._batch = self._batch, None
        if batch is None:
            return
        self.bwd.lane = 0
        for kind in (C.OP_POOL_REDUCE, C.OP_BN_BWD_REDUCE, C.OP_BN_BWD_FINALIZE, C.OP_GRAD_TERM):
            jobs = batch[kind]
            if not jobs:
                continue
            block = 0
            for op in jobs:
                if kind == C.OP_BN_BWD_FINALIZE:
                    nb = C.call('hrnet_ew_table_blocks', kind, self.dtid, 1, 1, 1, op.i[1])
                elif kind == C.OP_POOL_REDUCE:
                    nb = C.call('hrnet_ew_table_blocks', kind, self.dtid, op.i[1], op.i[2] >> 1, op.i[3] >> 1, op.i[4])
                else:
                    nb = C.call('hrnet_ew_table_blocks', kind, self.dtid, op.i[1], op.i[2], op.i[3], op.i[4])
                op.i[16], op.i[17] = block, nb
                block += nb
            arr = (C.HrOp * len(jobs))(*jobs)
            raw = bytes(ctypes.string_at(ctypes.addressof(arr), ctypes.sizeof(arr)))
            table = torch.frombuffer(bytearray(raw), dtype=torch.uint8).to(self.dev)
            self.keep.append(table)
            self.bwd.add(C.OP_EW_TABLE, ints=(len(jobs), block, kind, self.dtid), ptrs=(C.ptr(table),))
            self.n_batched_jobs += len(jobs)

    def _build_backward(self):
        net = self.net
        # d(heatmaps) NCHW f32 -> NHWC grad of the final conv; d(inter_feat) likewise (optional)
        oa = self.out_act
        self.gout_op = self.bwd.add(C.OP_NCHW_TO_NHWC, ints=(self.dtid, oa.N, oa.H, oa.W, oa.C, self.nj),
                                    ptrs=(None, C.ptr(oa.g)))
        oa.ginit = True
        # Weight gradients by float atomics (default): every workgroup of a weight-gradient / fused backward launch ADDS
        # its tile straight into the OIHW f32 gradient (3x3 tiles go through LDS so that a wave instruction covers 64
        # consecutive floats) instead of writing a slab per split that a reduce launch reads back: no slabs (1.94 GB
        # written and read back per step before), no reduce launches, and nothing left to do when the last
        # weight-gradient launch ends. HRNET_DETERMINISTIC=1 (or HRNET_WGRAD_ATOMIC=0) keeps slabs + ordered sums
        # (HRNET_WGRAD_ATOMIC=1 beside HRNET_DETERMINISTIC=1: ordered batch statistics, atomic weight gradients - tests).
        det = _knob('HRNET_DETERMINISTIC', '0') == '1'
        self.wgrad_atomic = (_knob('HRNET_BATCH_WRED', '1') != '0'
                             and _knob('HRNET_WGRAD_ATOMIC', '0' if det else '1') != '0')
        # batched backward of the fuse sums (HR_OP_EW_TABLE; HRNET_BATCH_SUMBWD=0: one launch per pass and lane)
        self.batch_sums = _knob('HRNET_BATCH_SUMBWD', '1') != '0'
        self._batch = None
        self.n_batched_jobs = 0
        relu_of = {}   # act -> relu flag its consumers apply (uniform per act in this network)
        for e in self.tape:
            if e[0] == 'conv':
                relu_of[id(e[1].act)] = e[1].relu
            elif e[0] == 'sum':
                for t in e[1]:
                    relu_of[id(t.act)] = t.relu
        # dgrad epilogue fusion of the BN-backward reduction: the dgrad conv that writes the LAST
        # contribution to an activation gradient gathers (sum dz, sum dz*y) of the BatchNorm behind it.
        first_use, producer_sum, use_lanes = {}, {}, {}
        for ti, (e, ln) in enumerate(zip(self.tape, self.tape_lanes)):
            if e[0] == 'conv':
                ins = [e[1].act]
            elif e[0] == 'sum':
                ins = [t.act for t in e[1]]
                producer_sum[id(e[4])] = e
            elif e[0] in ('cat', 'headmix'):
                ins = [v.act for v in e[1]]
            else:
                continue
            for a in ins:
                first_use.setdefault(id(a), ti)
                use_lanes.setdefault(id(a), set()).add(ln)
        # weight-gradient slab sums are batched: every layer keeps its own slab region and ONE table-driven
        # launch per lane segment (or per ~16 MB of gradient on lane 0) sums them
        self.batch_wred = _knob('HRNET_BATCH_WRED', '1') != '0'
        self._wred = {}            # lane -> pending (slabs tensor, HrWredEnt fields)
        self._wred_tables = []     # (op index, [entries]) patched with the device table at the end
        self._wred_bytes = 0
        self.slab_bytes = 0
        fuse_stats = self.training and _knob('HRNET_FUSE_BWDSTATS', '1') != '0'
        self.n_fused_bwdstats = 0
        self._bnrefs = []
        self.n_inline_bnbwd = 0
        self.inter_gop = None
        in_region = False
        self._producer_sum = producer_sum
        # Deferred weight gradients: the weight-gradient launches of the HighResolutionModules are off the critical
        # path, and the end of the backward pass (transition1, layer1, the stem: one lane, a third of its wall
        # time) leaves the other lanes idle - so they are recorded when their operands are final but ENQUEUED on
        # the side lanes when the walk reaches that single-lane tail. Their gradients are complete only at the
        # end of the program, so this is off under data parallelism (the bucketed exchange relies on gradients
        # completing in reverse layer order); HRNET_DEFER_WGRAD=0 turns it off.
        # Data parallelism (recorded at plan build: hipnet.optim.GradSync checks it): only the convolutions of the flat
        # buffer's LATE region (HipNet._flatten) are deferred - the bucketed exchange relies on every other gradient
        # completing in reverse layer order, and exchanges the late region when the program ends - and nothing is
        # offloaded from lane 0. HRNET_DP_PLAN=1 records that form in a single process (bench.py: what the step of
        # a data-parallel rank costs on one GPU).
        dp = (torch.distributed.is_available() and torch.distributed.is_initialized()
              and torch.distributed.get_world_size() > 1) or _knob('HRNET_DP_PLAN', '0') == '1'
        self.dp_plan = dp
        self.defer_wgrad = (self.nlanes > 1 and self.batch_wred and not self.wlane
                            and _knob('HRNET_DEFER_WGRAD', '1') != '0')
        self.defer_branch_wgrads = _knob('HRNET_DEFER_BRANCH', '1') != '0'   # (measurement: fuse layers only)
        # how much weight-gradient work the single-lane tail can hide: the tail is a stream over the stem / layer1
        # maps, so its length goes with their pixel count; w32 at B=64 needs 3.2 MFLOP per tail pixel to defer everything
        # (773 GFLOP behind a 4 ms tail). w48 has 1.9x the work per tail pixel: part of it stays in the modules
        # (deferring all of it: 43.8 ms/step, none: 37.0 in round 2; round 4, with the 96-channel branch on the fused
        # backward: 2.4 -> 29.1, 3.2 -> 28.8, 4.5 -> 28.4, 6 -> 28.4 ms/step; w32 unchanged)
        tail_pixels = self.N * (self.H // 4) * (self.W // 4)
        self._defer_budget = float(_knob('HRNET_DEFER_MFLOP_PER_PIXEL', '4.5')) * 1e6 * tail_pixels
        self._defer_flops = 0.0
        self.offload_wgrad = self.defer_wgrad and not dp and _knob('HRNET_OFFLOAD_WGRAD', '1') != '0'
        self._offload_rr = 0
        self._offload_lanes = set()
        self._deferred = []
        self._deferred_lanes = []
        self.late_cuts = []        # (backward op index, flat lo, flat hi, lane): late-region groups final on that lane there
        self.n_deferred_wgrads = 0
        first_fork = next((i for i, e in enumerate(self.tape) if e[0] == 'fork'), None)
        fused_at, fused_skip = self._find_fused_blocks(), set()
        self.n_fused_blocks = len(fused_at)
        self._fused_out_ids = {id(self.tape[ti][4]) for ti in fused_at}     # block outputs a fused launch consumes
        for ti, (e, lane) in reversed(list(enumerate(zip(self.tape, self.tape_lanes)))):
            self.bwd.lane = lane
            if ti in fused_skip:
                continue
            if ti in fused_at:
                if e[4] is self.inter_act and self.inter_gop is None:
                    self.inter_gop = len(self.bwd)
                if fused_at[ti] == 'bottleneck':
                    self._fused_bottleneck_backward(ti, lane, in_region)
                    fused_skip.update((ti - 1, ti - 2, ti - 3))
                elif fused_at[ti] == 'bottleneck_ds':
                    self._fused_bottleneck_backward(ti, lane, in_region, downsample=True)
                    fused_skip.update((ti - 1, ti - 2, ti - 3, ti - 4))
                else:
                    self._fused_block_backward(ti, lane, in_region)
                    fused_skip.update((ti - 1, ti - 2))
                continue
            if e[0] in ('fork', 'join'):
                # a forward join is the backward fork of the same lanes, and vice versa
                for l in sorted(self._wred):
                    self._flush_wred(l)
                self.bwd.lane = 0
                if self.batch_sums and len(e) > 2 and e[2] == 'sums':
                    # the backward of the module's fuse sums (a dozen reduce / finalize / apply passes over tensors of
                    # a few MB, three or four per lane in a row) runs as three batched launches on lane 0 instead:
                    # no fork / join around it
                    if e[0] == 'join':
                        self._batch = {C.OP_POOL_REDUCE: [], C.OP_BN_BWD_REDUCE: [], C.OP_BN_BWD_FINALIZE: [], C.OP_GRAD_TERM: []}
                    else:
                        self._flush_batch()
                elif e[0] == 'join':
                    self.bwd.fork(e[1])
                else:
                    self.bwd.join(e[1])
                if e[0] == 'join':
                    pass
                else:
                    if self.defer_wgrad and ti == first_fork:
                        self._emit_deferred_wgrads()
                    if self.wlane:
                        self.bwd.sync(self.wlane, 0)
                    # every gradient of this module (and of everything after it) is complete here
                    nxt = next((t for t in self.tape[ti + 1:] if t[0] == 'conv'), None)
                    if nxt is not None and self.batch_wred:
                        self.bucket_marks.append((len(self.bwd), nxt[2].prefix))
                in_region = e[0] == 'join'
                continue
            if e[0] == 'cat':
                _, vals, cat, align = e
                if cat is self.inter_act and self.inter_gop is None:
                    self.inter_gop = len(self.bwd)     # optional external gradient of inter_feat joins here
                hs = [v.act.H for v in vals] + [0] * (4 - len(vals))
                ws = [v.act.W for v in vals] + [0] * (4 - len(vals))
                cs = [v.act.C for v in vals] + [0] * (4 - len(vals))
                assert not any(v.act.ginit for v in vals)
                self.bwd.add(C.OP_BILINEAR_CAT_BWD,
                             ints=[self.dtid, len(vals), cat.N, cat.H, cat.W] + hs + ws + cs + [0],
                             floats=(1.0 if align else 0.0,),
                             ptrs=[C.ptr(cat.g)] + [C.ptr(v.act.g) for v in vals])
                for v in vals:
                    v.act.ginit = True
            elif e[0] == 'headmix':
                self._head_mix_backward(e, lane, in_region, relu_of)
            elif e[0] == 'sum':
                _, terms, shifts, relu_out, out = e
                if out is self.inter_act and self.inter_gop is None:
                    # optional external gradient of inter_feat: slot patched at run time
                    self.inter_gop = len(self.bwd)
                if not out.ginit:
                    # no consumer produced a gradient (cannot happen for this network)
                    raise RuntimeError('no gradient reaches ' + out.name)
                mask = C.ptr(out.t) if relu_out else None
                fusable = [t for t, sh in zip(terms, shifts)
                           if t.act.bn is not None and t.act.nuse == 1 and sh == 0 and not t.relu]
                plain = [t for t, sh in zip(terms, shifts) if not (t.act.bn is not None and t.act.nuse == 1)]
                # one BN term and one identity/accumulating term share dz: written by the same pass
                paired = (fusable[0], plain[0]) if fusable and plain else (None, None)
                # the nearest-up-sampled terms (from the branches below this output): ONE walk over out.g and the mask
                # pools every level, stores each level's dz and gathers its sums (HR_OP_POOL_REDUCE)
                pooled = {}
                ups = sorted([(sh, t) for t, sh in zip(terms, shifts)
                              if sh > 0 and t.act.bn is not None and t.act.nuse == 1 and self.training and not t.relu
                              and t.act.bwd_rows is None], key=lambda q: q[0])
                if (ups and [q[0] for q in ups] == list(range(1, len(ups) + 1)) and len(ups) <= 3
                        and out.H % (1 << len(ups)) == 0 and out.W % (1 << len(ups)) == 0
                        and 256 // (out.C // (4 if self.dt == torch.float32 else 8)) >= 4 ** (len(ups) - 1)
                        and _knob('HRNET_POOL_REDUCE', '1') != '0' and _knob('HRNET_KEEP_DZ', '1') != '0'):
                    L = len(ups)
                    pblocks = C.call('hrnet_ew_table_blocks', C.OP_POOL_REDUCE, self.dtid, out.N, out.H >> 1, out.W >> 1, out.C)
                    ptrs = [C.ptr(out.g), mask]
                    for sh, t in ups:
                        part = self._f32(pblocks * 2 * out.C)
                        ptrs += [C.ptr(t.act.t), C.ptr(t.act.g), C.ptr(part)]
                        pooled[id(t.act)] = (part, pblocks)
                    self._emit(C.OP_POOL_REDUCE, ints=(self.dtid, out.N, out.H, out.W, out.C, L), ptrs=ptrs)
                for t, sh in zip(terms, shifts):
                    a = t.act
                    if t is paired[1]:
                        continue
                    if a.bn is not None and a.nuse == 1 and self.training:
                        # single consumer: fuse pooling + masks + BatchNorm backward
                        self._bn_backward(a, C.ptr(out.g), mask, sh, t.relu,
                                          extra=paired[1].act if t is paired[0] else None, pooled=pooled.get(id(a)))
                    else:
                        # accumulate d(post-activation value); BN backward runs at the producer
                        assert sh == 0
                        self._emit(C.OP_GRAD_TERM,
                                   ints=(self.dtid, a.N, a.H, a.W, a.C, 0, 0, 1 if a.ginit else 0),
                                   ptrs=(C.ptr(a.g), C.ptr(out.g), mask, None, None, None, None))
                        a.ginit = True
            elif e[0] == 'conv':
                _, xin, crec, stride, y, bnrec = e
                x = xin.act
                ks = 1 if crec.stem else crec.ks
                self.bwd.tags[len(self.bwd)] = crec.prefix
                if not y.ginit:
                    raise RuntimeError('no gradient reaches ' + y.name)
                if bnrec is not None and not y.bn_done:
                    # y.g holds d(post-activation) summed over consumers -> d(raw), in place
                    self._bn_backward(y, C.ptr(y.g), None, 0, relu_of.get(id(y), False))
                w = crec.mod.weight
                # a bias in front of a batch-statistics BatchNorm has an exactly zero gradient (the BatchNorm-backward
                # gradient sums to zero over every channel; autograd's value is rounding noise): no pass over dY
                if crec.mod.bias is not None and not (bnrec is not None and self.training):
                    blocks = C.call('hrnet_reduce_blocks', 1, 1, y.pixels, y.C)
                    self.max_bwd_part = max(self.max_bwd_part, blocks * y.C)
                    i = self.bwd.add(C.OP_BIAS_GRAD, ints=(self.dtid, y.pixels, y.C, crec.Cout, 1),
                                     ptrs=(C.ptr(y.g), C.ptr(net.grad_of(crec.mod.bias)), None))
                    self._scratch(self.bwd, i, 2, 'bwdpart')
                # the weight gradient is off the critical path: it runs on its own lane once dY is final
                if self.wlane and (self.wlane_all or lane == 0):
                    self.bwd.sync(lane, self.wlane)
                    self.bwd.lane = self.wlane
                nsplit = C.call('hrnet_wgrad_splits', self.dtid, x.N, y.H, y.W, y.C, x.C, ks, stride)
                wflops = 2.0 * x.N * y.H * y.W * y.C * x.C * ks * ks
                deferred = (self.batch_wred and self.defer_wgrad and in_region and first_fork is not None
                            and ti > first_fork
                            and (self.defer_branch_wgrads or '.branches.' not in crec.prefix)
                            and (not self.dp_plan or net.is_late(w))
                            and self._defer_flops + wflops <= self._defer_budget)
                if deferred:
                    self._defer_flops += wflops
                if deferred and nsplit > 1:
                    # A deferred launch runs in the background of the single-lane tail: it does not need the
                    # parallelism of many splits, and every split is a slab written and read back - but it must
                    # keep ~100 workgroups (the wide w48 layers have few splits to begin with: dividing those
                    # cost 6 ms per step). measured (w32 B=64, ms/step): divisor 1: 19.97, 2: 19.56, 4: 19.49, 8: 19.97
                    div = int(_knob('HRNET_DEFER_SPLIT_DIV', '4'))
                    tiles = C.call('hrnet_wgrad_tiles', self.dtid, x.N, y.H, y.W, y.C, x.C, ks, stride)
                    per = C.call('hrnet_wgrad_blocks_per_split', self.dtid, y.H, y.W, y.C, x.C, ks, stride)
                    floor_ = min(nsplit, -(-96 // max(per, 1)))
                    nsplit = max(1, floor_, nsplit // max(div, 1))
                    while nsplit > 1 and tiles % nsplit != 0:
                        nsplit -= 1
                # (the stem's weights are a flattened 3x3x3 kernel over im2col columns: its slab layout stays)
                direct = self.batch_wred and self.wgrad_atomic and not crec.stem
                wints = (self.dtid, x.N, x.H, x.W, x.C, y.H, y.W, y.C, ks, stride, 1 if xin.relu else 0, nsplit,
                         1 if direct else 0, crec.Cout, crec.Cin)
                wptrs = [C.ptr(x.t), C.ptr(y.g), C.ptr(xin.bn.scale) if xin.bn else None,
                         C.ptr(xin.bn.shift) if xin.bn else None, None]
                if direct:
                    wptrs[4] = C.ptr(net.grad_of(w))
                    if deferred:
                        self._deferred.append((wints, wptrs, None, 2.0 * x.N * y.H * y.W * y.C * x.C * ks * ks,
                                               net.offsets[id(w)][0]))
                    elif self.offload_wgrad and lane == 0 and not in_region:
                        l = 1 + self._offload_rr % max(1, self.nlanes - 1)
                        self._offload_rr += 1
                        self.bwd.sync(0, l)
                        self.bwd.add(C.OP_WGRAD, ints=wints, ptrs=wptrs, lane=l)
                        self._offload_lanes.add(l)
                    else:
                        self.bwd.add(C.OP_WGRAD, ints=wints, ptrs=wptrs)
                        if lane == 0:
                            self._wred_bytes += crec.Cout * crec.Cin * crec.ks * crec.ks * 4
                elif self.batch_wred:
                    slabs = self._f32(nsplit * y.C * ks * ks * x.C)
                    self.slab_bytes += slabs.numel() * 4
                    wptrs[4] = C.ptr(slabs)
                    ent = dict(slabs=C.ptr(slabs), grad=C.ptr(net.grad_of(w)), nsplit=nsplit, Cout_pad=y.C, Cin_pad=x.C,
                               ks=crec.ks if crec.stem else ks, Cout=crec.Cout, Cin=crec.Cin, kflat=1 if crec.stem else 0,
                               accumulate=1)
                    if deferred:
                        # x.t, y.g and the BatchNorm coefficients of xin stay untouched until the program ends
                        self._deferred.append((wints, wptrs, ent, 2.0 * x.N * y.H * y.W * y.C * x.C * ks * ks,
                                               net.offsets[id(w)][0]))
                    elif self.offload_wgrad and lane == 0 and not in_region:
                        # a single-lane part of the pass (head, transition1, stem): its weight gradients are off the
                        # dependency chain - they go to a side lane, which idles there (the head) or carries the
                        # deferred launches (the tail); their gradient is complete at the end of the program
                        l = 1 + self._offload_rr % max(1, self.nlanes - 1)
                        self._offload_rr += 1
                        self.bwd.sync(0, l)
                        self.bwd.add(C.OP_WGRAD, ints=wints, ptrs=wptrs, lane=l)
                        self._wred.setdefault(l, []).append(ent)
                        self._offload_lanes.add(l)
                    else:
                        self.bwd.add(C.OP_WGRAD, ints=wints, ptrs=wptrs)
                        self._wred.setdefault(self.bwd.lane, []).append(ent)
                        if lane == 0:
                            self._wred_bytes += crec.Cout * crec.Cin * crec.ks * crec.ks * 4
                else:
                    self.max_slab = max(self.max_slab, nsplit * y.C * ks * ks * x.C)
                    i = self.bwd.add(C.OP_WGRAD, ints=wints, ptrs=wptrs)
                    self._scratch(self.bwd, i, 4, 'slab')
                    i = self.bwd.add(C.OP_WGRAD_REDUCE,
                                     ints=(nsplit, y.C, x.C, ks, crec.Cout, crec.Cin, 1 if crec.stem else 0, 1),
                                     ptrs=(None, C.ptr(net.grad_of(w))))
                    self._scratch(self.bwd, i, 0, 'slab')
                    if crec.stem:
                        self.bwd.ops[i].i[3] = crec.ks   # real taps of the flattened stem kernel
                self.bwd.lane = lane
                if (x is self._head_y and x.g is not None and xin.bn is not None and self.training and ks == 1
                        and crec.mod.weight.shape[0] <= y.C and _knob('HRNET_HEAD_BWD', '1') != '0'
                        and C.call('hrnet_head_mix_supported', self.dtid, y.C, x.C) == 1):
                    # the layer behind the head's BatchNorm (last_layer.3): its input gradient dz = W^T dY is a K = 32
                    # product per pixel - formed here only to gather the BatchNorm-backward sums, and formed AGAIN by
                    # the apply launch in _head_mix_backward (hrnet_head_bwd): dz is never stored or re-read
                    nrows = C.call('hrnet_head_mix_rows', x.N, x.H, x.W)
                    rows = self._f32(nrows * 2 * x.C)
                    self.bwd.add(C.OP_HEAD_BWD, ints=(self.dtid, x.N, x.H, x.W, y.C, x.C, 1, 1 if xin.relu else 0),
                                 ptrs=(C.ptr(y.g), C.ptr(crec.wd), C.ptr(x.t), C.ptr(rows), C.ptr(xin.bn.scale),
                                       C.ptr(xin.bn.shift), None))
                    x.bwd_rows = (rows, nrows)
                    x.ginit = True
                    self._head_bwd = (y, crec, 1 if xin.relu else 0)
                    self.n_fused_bwdstats += 1
                elif x.g is not None:
                    # input gradient = conv of dY with the transposed kernel (zero-stuffed for stride 2)
                    ptrs = [C.ptr(y.g), C.ptr(crec.wd), None, None, None, C.ptr(x.g), None, None, None, None, None]
                    last = (first_use.get(id(x)) == ti and use_lanes.get(id(x)) == {lane}
                            and x is not self.inter_act)
                    target = None
                    if fuse_stats and last and xin.bn is not None and x.nuse == 1:
                        # x is a raw conv output read through its BatchNorm (+ReLU) by this conv alone
                        target = x
                        ptrs[7] = C.ptr(x.t)
                        if xin.relu:
                            ptrs[9], ptrs[10] = C.ptr(xin.bn.scale), C.ptr(xin.bn.shift)
                    elif fuse_stats and last and xin.bn is None and id(x) in producer_sum:
                        # x is the output of a sum: gather for its first plain BatchNorm term
                        _, terms, shifts, relu_out, _o = producer_sum[id(x)]
                        cand = [t for t, sh in zip(terms, shifts)
                                if t.act.bn is not None and t.act.nuse == 1 and sh == 0 and not t.relu]
                        if cand:
                            target = cand[0].act
                            ptrs[7] = C.ptr(target.t)
                            ptrs[8] = C.ptr(x.t) if relu_out else None
                    store_masked = 0
                    if target is not None:
                        nrows = C.call('hrnet_conv_rows_bwdstats', self.dtid, x.N, x.H, x.W, y.C, x.C, ks, stride)
                        rows = self._f32(nrows * 2 * x.C)
                        ptrs[6] = C.ptr(rows)
                        target.bwd_rows = (rows, nrows)
                        self.n_fused_bwdstats += 1
                        if (target is not x and ptrs[8] is not None and id(x) in self._fused_out_ids
                                and _knob('HRNET_BS_STORE_MASKED', '1') != '0'):
                            # x closes a block whose backward is a fused launch: this (last) contribution stores the
                            # gradient already multiplied by x's ReLU mask - no separate mask pass over the tensor
                            store_masked = 1
                    # (a backward-statistics launch is bound to the kernel family its rows buffer was sized for)
                    route = (C.call('hrnet_conv_route', self.dtid, x.N, x.H, x.W, y.C, x.C, ks, stride)
                             if target is not None else 0)
                    self.bwd.add(C.OP_CONV,
                                 ints=(self.dtid, y.N, y.H, y.W, y.C, x.H, x.W, x.C, ks, stride,
                                       1 if stride == 2 else 0, 0, 1 if x.ginit else 0, 0, store_masked, 0, 0, route),
                                 ptrs=ptrs)
                    x.ginit = True
                    if store_masked:
                        x.gmasked = True
                if lane == 0 and not in_region:
                    self._bucket_mark_after_conv(crec)
        if self.batch_wred:
            for l in sorted(self._wred):
                self._flush_wred(l)
            self.bwd.lane = 0
            late = sorted(set(self._deferred_lanes) | self._offload_lanes)
            if late:
                self.bwd.join(late)       # the deferred / offloaded weight gradients (and their slab sums) are done
            self._upload_wred_tables()

    # ---- backward of the head without its concat (head_mix) ----
    def _head_mix_backward(self, e, lane, in_region, relu_of):
        """G = d(raw y) (BatchNorm backward in place); g_j = up^T(G) at branch j's resolution (hrnet_upsample_bilinear_t);
        dW[:, slice j] = g_j^T x_j and dx_j = W_j^T g_j as 1x1 launches at branch resolution (g_0 = G, full resolution
        with K = C0): autograd of pose_hrnet.py:560-566, an eighth of the concat form's FLOPs"""
        _, vals, crec, y, bnrec, ts, offs, align = e
        net = self.net
        self.bwd.tags[len(self.bwd)] = crec.prefix
        if not y.ginit:
            raise RuntimeError('no gradient reaches ' + y.name)
        if bnrec is not None and not y.bn_done and self._head_bwd is not None:
            # G = A*dz + B*y + C with dz = W3^T dHM formed again from the next layer's gradient (hrnet_head_bwd mode 2)
            ny, ncrec, relu_flag = self._head_bwd
            b, m = y.bn, y.bn.mod
            rows, nrows = y.bwd_rows
            self.bwd.add(C.OP_BN_BWD_FINALIZE, ints=(nrows, y.C, 1), floats=(y.pixels,),
                         ptrs=(C.ptr(rows), C.ptr(m.weight), C.ptr(b.mean), C.ptr(b.invstd),
                               C.ptr(net.grad_of(m.weight)), C.ptr(net.grad_of(m.bias)), C.ptr(b.coef)))
            self.bwd.add(C.OP_HEAD_BWD, ints=(self.dtid, y.N, y.H, y.W, ny.C, y.C, 2, relu_flag),
                         ptrs=(C.ptr(ny.g), C.ptr(ncrec.wd), C.ptr(y.t), C.ptr(y.g), C.ptr(b.scale), C.ptr(b.shift),
                               C.ptr(b.coef)))
            y.bn_done = True
        elif bnrec is not None and not y.bn_done:
            self._bn_backward(y, C.ptr(y.g), None, 0, relu_of.get(id(y), False))
        w = crec.mod.weight
        if crec.mod.bias is not None and not (bnrec is not None and self.training):
            blocks = C.call('hrnet_reduce_blocks', 1, 1, y.pixels, y.C)
            self.max_bwd_part = max(self.max_bwd_part, blocks * y.C)
            i = self.bwd.add(C.OP_BIAS_GRAD, ints=(self.dtid, y.pixels, y.C, crec.Cout, 1),
                             ptrs=(C.ptr(y.g), C.ptr(net.grad_of(crec.mod.bias)), None))
            self._scratch(self.bwd, i, 2, 'bwdpart')
        if ts:
            ints = [self.dtid, y.N, y.H, y.W, y.C, len(ts), 1 if align else 0]
            for t in ts:
                ints += [t.H, t.W]
                t.ginit = True
            self.bwd.add(C.OP_UPSAMPLE_T, ints=ints, ptrs=[C.ptr(y.g)] + [C.ptr(t.g) for t in ts])
        gw = net.grad_of(w)
        direct = self.batch_wred and self.wgrad_atomic
        for j, v in enumerate(vals):
            x = v.act
            dy = y.g if j == 0 else ts[j - 1].g
            cj = offs[j + 1] - offs[j]
            gptr = gw.data_ptr() + 4 * offs[j]
            # ---- weight gradient of the column slice (rows crec.Cin floats apart)
            nsplit = C.call('hrnet_wgrad_splits', self.dtid, x.N, x.H, x.W, y.C, x.C, 1, 1)
            wints = [self.dtid, x.N, x.H, x.W, x.C, x.H, x.W, y.C, 1, 1, 0, nsplit, 1 if direct else 0, crec.Cout, cj,
                     crec.Cin if direct else 0]
            wptrs = [C.ptr(x.t), C.ptr(dy), None, None, None]
            side = None
            if self.offload_wgrad and lane == 0 and not in_region:
                side = 1 + self._offload_rr % max(1, self.nlanes - 1)
                self._offload_rr += 1
                self.bwd.sync(0, side)
                self._offload_lanes.add(side)
            if direct:
                wptrs[4] = gptr
                self.bwd.add(C.OP_WGRAD, ints=wints, ptrs=wptrs, lane=side)
            elif self.batch_wred:
                slabs = self._f32(nsplit * y.C * x.C)
                self.slab_bytes += slabs.numel() * 4
                wptrs[4] = C.ptr(slabs)
                ent = dict(slabs=C.ptr(slabs), grad=gptr, nsplit=nsplit, Cout_pad=y.C, Cin_pad=x.C, ks=1,
                           Cout=crec.Cout, Cin=cj, kflat=0, accumulate=1, ld=crec.Cin)
                self.bwd.add(C.OP_WGRAD, ints=wints, ptrs=wptrs, lane=side)
                self._wred.setdefault(side if side is not None else self.bwd.lane, []).append(ent)
            else:
                self.max_slab = max(self.max_slab, nsplit * y.C * x.C)
                i = self.bwd.add(C.OP_WGRAD, ints=wints, ptrs=wptrs)
                self._scratch(self.bwd, i, 4, 'slab')
                i = self.bwd.add(C.OP_WGRAD_REDUCE, ints=(nsplit, y.C, x.C, 1, crec.Cout, cj, 0, 1, crec.Cin),
                                 ptrs=(None, gptr))
                self._scratch(self.bwd, i, 0, 'slab')
            if side is None and lane == 0:
                self._wred_bytes += crec.Cout * cj * 4
            # ---- input gradient: rows offs[j] .. offs[j+1] of the transposed packed weight ([Cin][Cout_pad])
            if x.g is not None:
                wd = crec.wd.data_ptr() + offs[j] * crec.Cout_pad * self.esize
                self.bwd.add(C.OP_CONV,
                             ints=(self.dtid, x.N, x.H, x.W, y.C, x.H, x.W, x.C, 1, 1, 0, 0, 1 if x.ginit else 0, 0, 0),
                             ptrs=[C.ptr(dy), wd, None, None, None, C.ptr(x.g), None, None, None, None, None])
                x.ginit = True
        if lane == 0 and not in_region:
            self._bucket_mark_after_conv(crec)

    # ---- fused backward of a BasicBlock (conv3x3+BN+ReLU, conv3x3+BN, +x, ReLU: pose_hrnet.py:41-57) ----
    def _find_fused_blocks(self):
        """tape indices of the 'sum' entries that close a BasicBlock (or an identity Bottleneck) all of whose convs
        the fused kernels serve -> 'basic' / 'bottleneck'"""
        out = {}
        if not self.training or _knob('HRNET_FUSED_BWD', '1') == '0':
            return out
        T, L = self.tape, self.tape_lanes
        # Widths above 64 (the 128-channel instantiation): only where the LDS-ring pipeline does NOT serve the block's
        # input-gradient convs. w32's 128-channel branch (16x16 maps) has the ring: fused loses there, 19.32 vs 18.82 ms -
        # its four input-channel blocks re-stage the same 128-channel g tile. w48's 96-channel branch (48x36 maps) has
        # only the tile-walking body: fused wins, 28.72 vs 29.92 ms/step (round 4).
        maxc = int(_knob('HRNET_FUSED_MAXC', '128'))     # (tests: restrict the fused path to narrow layers)

        def wide_ok(x, y1, y2):
            if max(x.C, y1.C, y2.C) <= 64:
                return True
            return (C.call('hrnet_conv_route', self.dtid, x.N, x.H, x.W, y1.C, x.C, 3, 1) != 2
                    and C.call('hrnet_conv_route', self.dtid, x.N, x.H, x.W, y2.C, y1.C, 3, 1) != 2)
        for ti in range(2, len(T)):
            e = T[ti]
            if e[0] != 'sum' or T[ti - 1][0] != 'conv' or T[ti - 2][0] != 'conv':
                continue
            _, terms, shifts, relu_out, res = e
            _, xin2, crec2, st2, y2, bn2 = T[ti - 1]
            _, xin1, crec1, st1, y1, bn1 = T[ti - 2]
            if not (len(terms) == 2 and list(shifts) == [0, 0] and relu_out and L[ti] == L[ti - 1] == L[ti - 2]):
                continue
            c, idt = terms
            x = xin1.act
            ok = (c.act is y2 and c.bn is bn2 and bn2 is not None and not c.relu and y2.nuse == 1
                  and xin2.act is y1 and xin2.bn is bn1 and bn1 is not None and xin2.relu and y1.nuse == 1
                  and idt.act is x and idt.bn is xin1.bn and idt.relu == xin1.relu and x.nuse == 2
                  and st1 == 1 and st2 == 1 and crec1.ks == 3 and crec2.ks == 3 and not crec1.stem
                  and crec1.mod.bias is None and crec2.mod.bias is None and x.g is not None
                  and max(x.C, y1.C, y2.C) <= maxc and wide_ok(x, y1, y2)
                  and C.call('hrnet_bwd_fused_supported', self.dtid, x.C, y1.C)
                  and C.call('hrnet_bwd_fused_supported', self.dtid, y1.C, y2.C))
            # the mask the kernel applies to the gradient it stores for x is [a > 0], a = the conv's input as staged:
            # right when x is read through a ReLU, or is the (non-negative) output of a sum that ended in one
            ps = self._producer_sum.get(id(x))
            ok = ok and (xin1.relu or (xin1.bn is None and ps is not None and ps[3]))
            if ok:
                out[ti] = 'basic'
        # Bottleneck with an identity residual (pose_hrnet.py:60-105; layer1 blocks 1..3): conv1 1x1, conv2 3x3,
        # conv3 1x1, + x, ReLU - the pointwise convs through hrnet_conv1x1_bwd_fused
        if _knob('HRNET_FUSED_PW', '1') != '0':
            for ti in range(3, len(T)):
                e = T[ti]
                if e[0] != 'sum' or any(T[ti - k][0] != 'conv' for k in (1, 2, 3)):
                    continue
                _, terms, shifts, relu_out, res = e
                _, xin3, crec3, st3, y3, bn3 = T[ti - 1]
                _, xin2, crec2, st2, y2, bn2 = T[ti - 2]
                _, xin1, crec1, st1, y1, bn1 = T[ti - 3]
                if not (len(terms) == 2 and list(shifts) == [0, 0] and relu_out
                        and L[ti] == L[ti - 1] == L[ti - 2] == L[ti - 3]):
                    continue
                c, idt = terms
                x = xin1.act
                ps = self._producer_sum.get(id(x))
                ok = (c.act is y3 and c.bn is bn3 and bn3 is not None and not c.relu and y3.nuse == 1
                      and xin3.act is y2 and xin3.bn is bn2 and bn2 is not None and xin3.relu and y2.nuse == 1
                      and xin2.act is y1 and xin2.bn is bn1 and bn1 is not None and xin2.relu and y1.nuse == 1
                      and idt.act is x and idt.bn is None and not idt.relu and xin1.bn is None and x.nuse == 2
                      and ps is not None and ps[3]
                      and st1 == st2 == st3 == 1 and (crec1.ks, crec2.ks, crec3.ks) == (1, 3, 1)
                      and not (crec1.stem or crec2.stem or crec3.stem)
                      and crec1.mod.bias is None and crec2.mod.bias is None and crec3.mod.bias is None
                      and x.g is not None and x is not self.inter_act
                      and C.call('hrnet_bwd_pw_supported', self.dtid, y2.C, y3.C)
                      and C.call('hrnet_bwd_fused_supported', self.dtid, y1.C, y2.C)
                      and C.call('hrnet_bwd_pw_supported', self.dtid, x.C, y1.C))
                if ok:
                    out[ti] = 'bottleneck'
            # ... and the first Bottleneck of layer1, whose residual is a 1x1 conv + BatchNorm of the same input
            # (pose_hrnet.py:98-99: downsample): conv1, conv2, conv3, downsample, sum
            for ti in range(4, len(T)):
                e = T[ti]
                if e[0] != 'sum' or any(T[ti - k][0] != 'conv' for k in (1, 2, 3, 4)):
                    continue
                _, terms, shifts, relu_out, res = e
                _, xind, crecd, std, yd, bnd = T[ti - 1]
                _, xin3, crec3, st3, y3, bn3 = T[ti - 2]
                _, xin2, crec2, st2, y2, bn2 = T[ti - 3]
                _, xin1, crec1, st1, y1, bn1 = T[ti - 4]
                if not (len(terms) == 2 and list(shifts) == [0, 0] and relu_out
                        and len({L[ti - k] for k in range(5)}) == 1):
                    continue
                c, d = terms
                x = xin1.act
                ok = (c.act is y3 and c.bn is bn3 and bn3 is not None and not c.relu and y3.nuse == 1
                      and d.act is yd and d.bn is bnd and bnd is not None and not d.relu and yd.nuse == 1
                      and xin3.act is y2 and xin3.bn is bn2 and bn2 is not None and xin3.relu and y2.nuse == 1
                      and xin2.act is y1 and xin2.bn is bn1 and bn1 is not None and xin2.relu and y1.nuse == 1
                      and xind.act is x and xind.bn is xin1.bn and xind.relu == xin1.relu and xin1.relu
                      and xin1.bn is not None and x.nuse == 2
                      and st1 == st2 == st3 == std == 1 and (crec1.ks, crec2.ks, crec3.ks, crecd.ks) == (1, 3, 1, 1)
                      and not (crec1.stem or crec2.stem or crec3.stem or crecd.stem)
                      and all(cr.mod.bias is None for cr in (crec1, crec2, crec3, crecd))
                      and x.g is not None and x is not self.inter_act
                      and C.call('hrnet_bwd_pw_supported', self.dtid, y2.C, y3.C)
                      and C.call('hrnet_bwd_pw_supported', self.dtid, x.C, yd.C)
                      and C.call('hrnet_bwd_fused_supported', self.dtid, y1.C, y2.C)
                      and C.call('hrnet_bwd_pw_supported', self.dtid, x.C, y1.C))
                if ok:
                    out[ti] = 'bottleneck_ds'
        return out

    def _bn_bwd_finalize(self, y, reduce_from=None):
        """coefficients (and dgamma/dbeta) of y's BatchNorm backward from the sums a previous launch left in
        y.bwd_rows, or from a reduction pass over the (already masked) gradient `reduce_from`"""
        b, m = y.bn, y.bn.mod
        tail = (C.ptr(m.weight), C.ptr(b.mean), C.ptr(b.invstd), C.ptr(self.net.grad_of(m.weight)),
                C.ptr(self.net.grad_of(m.bias)), C.ptr(b.coef))
        if y.bwd_rows is not None:
            rows, blocks = y.bwd_rows
            self.bwd.add(C.OP_BN_BWD_FINALIZE, ints=(blocks, y.C, 1), floats=(y.pixels,), ptrs=(C.ptr(rows),) + tail)
        else:
            blocks = C.call('hrnet_reduce_blocks', y.N, y.H, y.W, y.C)
            self.max_bwd_part = max(self.max_bwd_part, blocks * 2 * y.C)
            i = self.bwd.add(C.OP_BN_BWD_REDUCE, ints=(self.dtid, y.N, y.H, y.W, y.C, 0, 0),
                             ptrs=(None, reduce_from, None, C.ptr(y.t), None, None))
            self._scratch(self.bwd, i, 0, 'bwdpart')
            j = self.bwd.add(C.OP_BN_BWD_FINALIZE, ints=(blocks, y.C, 1), floats=(y.pixels,), ptrs=(None,) + tail)
            self._scratch(self.bwd, j, 0, 'bwdpart')
        y.bn_done = True

    def _fused_conv_bwd(self, dz, y, xin, crec, dx, addend, mask_out, rows_for, lane, reduce_from=None):
        """one hrnet_conv3x3_bwd_fused / hrnet_conv1x1_bwd_fused launch: backward of `y = conv(xin)` given the masked
        gradient `dz` of y's BatchNorm output; writes the gradient of xin's activation into `dx` and y's
        weight-gradient slabs. rows_for: the raw activation whose BatchNorm-backward sums the launch gathers.
        The BatchNorm backward of y itself is finished first: by the launch (from the partial rows a previous
        launch left: HrBnBwdRef, no finalize launch in between) when there are few enough rows, else by
        _bn_bwd_finalize (rows, or a reduction pass over `reduce_from`)."""
        net, x = self.net, xin.act
        ks = crec.ks
        ref = None
        lim = 16384 if ks == 3 else 8192
        if (y.bwd_rows is not None and y.bwd_rows[1] * y.C <= lim
                and _knob('HRNET_BNBWD_INLINE', '1') != '0'):
            b, m = y.bn, y.bn.mod
            ref = C.HrBnBwdRef()
            ref.rows, ref.gamma = C.ptr(y.bwd_rows[0]), C.ptr(m.weight)
            ref.save_mean, ref.save_invstd = C.ptr(b.mean), C.ptr(b.invstd)
            ref.dgamma, ref.dbeta = C.ptr(self.net.grad_of(m.weight)), C.ptr(self.net.grad_of(m.bias))
            ref.count, ref.nrows, ref.accumulate = float(y.pixels), int(y.bwd_rows[1]), 1
            self._bnrefs.append(ref)             # host structs the recorded op points at
            self.n_inline_bnbwd += 1
            y.bn_done = True
        else:
            self._bn_bwd_finalize(y, reduce_from=reduce_from)
        if ks == 3:
            ns = C.call('hrnet_bwd_fused_splits', self.dtid, x.N, x.H, x.W, x.C, y.C)
            kind = C.OP_BWD_FUSED
        else:
            ns = C.call('hrnet_bwd_pw_splits', self.dtid, x.pixels, x.C, y.C)
            kind = C.OP_BWD_PW
            if rows_for is not None and not C.call('hrnet_bwd_pw_rows_supported', self.dtid, x.C, y.C):
                rows_for = None               # (the BatchNorm backward behind it runs its own reduction pass)
        w = crec.mod.weight
        atomic = 1 if (self.batch_wred and self.wgrad_atomic
                       and (ks == 3 or (crec.Cout == y.C and crec.Cin == x.C))) else 0
        if atomic:
            slabs = net.grad_of(w)               # the launch adds into the OIHW gradient itself
        else:
            slabs = self._f32(ns * y.C * ks * ks * x.C)
            self.slab_bytes += slabs.numel() * 4
        rows = None
        if rows_for is not None:
            rows = self._f32(ns * 2 * x.C)
            rows_for.bwd_rows = (rows, ns)
            self.n_fused_bwdstats += 1
        self.bwd.add(kind,
                     ints=(self.dtid, x.N, x.H, x.W, x.C, y.C, 1 if xin.relu else 0, 1 if mask_out else 0, atomic,
                           crec.Cout, crec.Cin),
                     ptrs=(dz, C.ptr(y.t), None if ref is not None else C.ptr(y.bn.coef), C.ptr(x.t),
                           C.ptr(xin.bn.scale) if xin.bn else None, C.ptr(xin.bn.shift) if xin.bn else None,
                           C.ptr(crec.wd), dx, addend, C.ptr(rows),
                           C.ptr(rows_for.t) if rows_for is not None else None, C.ptr(slabs),
                           ctypes.addressof(ref) if ref is not None else None))
        if atomic:
            if lane == 0:
                self._wred_bytes += crec.Cout * crec.Cin * ks * ks * 4
            return
        ent = dict(slabs=C.ptr(slabs), grad=C.ptr(net.grad_of(w)), nsplit=ns, Cout_pad=y.C, Cin_pad=x.C, ks=ks,
                   Cout=crec.Cout, Cin=crec.Cin, kflat=0, accumulate=1)
        if self.batch_wred:
            self._wred.setdefault(lane, []).append(ent)
            if lane == 0:
                self._wred_bytes += crec.Cout * crec.Cin * ks * ks * 4
        else:
            self.bwd.add(C.OP_WGRAD_REDUCE, ints=(ns, y.C, x.C, ks, crec.Cout, crec.Cin, 0, 1),
                         ptrs=(C.ptr(slabs), C.ptr(net.grad_of(w))))

    def _rows_target(self, x, xin):
        """the raw activation behind x whose BatchNorm backward needs (sum dz, sum dz*y) of x's gradient"""
        if x is self.inter_act:
            return None
        if xin.bn is not None:
            return x                             # x is a raw conv output read through its BatchNorm + ReLU
        ps = self._producer_sum.get(id(x))
        cand = [t for t, sh in zip(ps[1], ps[2])
                if t.act.bn is not None and t.act.nuse == 1 and sh == 0 and not t.relu] if ps else []
        return cand[0].act if cand else None

    def _fused_bottleneck_backward(self, ti, lane, in_region, downsample=False):
        """Bottleneck: fused launches for conv3 1x1, conv2 3x3, conv1 1x1 (+ the residual stream: the identity's
        masked gradient, or the input gradient of the downsample conv written first)"""
        _, terms, shifts, relu_out, out = self.tape[ti]
        k = 1 if downsample else 0
        _, xin3, crec3, _, y3, bn3 = self.tape[ti - 1 - k]
        _, xin2, crec2, _, y2, bn2 = self.tape[ti - 2 - k]
        _, xin1, crec1, _, y1, bn1 = self.tape[ti - 3 - k]
        x = xin1.act
        self.bwd.tags[len(self.bwd)] = crec3.prefix
        if not out.ginit:
            raise RuntimeError('no gradient reaches ' + out.name)
        assert not x.ginit and not y1.ginit and not y2.ginit and not y3.ginit
        if not out.gmasked:
            self.bwd.add(C.OP_GRAD_TERM, ints=(self.dtid, out.N, out.H, out.W, out.C, 0, 0, 0, 0),
                         ptrs=(C.ptr(out.g), C.ptr(out.g), C.ptr(out.t), None, None, None, None, None))
            out.gmasked = True
        residual = C.ptr(out.g)              # identity: the masked d(out) joins the input gradient of conv1
        if downsample:
            _, xind, crecd, _, yd, bnd = self.tape[ti - 1]
            assert not yd.ginit
            self.bwd.tags[len(self.bwd)] = crecd.prefix
            self._fused_conv_bwd(C.ptr(out.g), yd, xind, crecd, C.ptr(x.g), None, True, None, lane,
                                 reduce_from=C.ptr(out.g))
            yd.ginit = True
            residual = C.ptr(x.g)            # conv1's launch adds its own input gradient in place
            self.bwd.tags[len(self.bwd)] = crec3.prefix
        # conv3: dz = d(out) masked; its input is relu(bn2(y2))
        self._fused_conv_bwd(C.ptr(out.g), y3, xin3, crec3, C.ptr(y2.g), None, True, y2, lane,
                             reduce_from=C.ptr(out.g))
        y2.ginit = y2.gmasked = True
        y3.ginit = True
        self.bwd.tags[len(self.bwd)] = crec2.prefix
        self._fused_conv_bwd(C.ptr(y2.g), y2, xin2, crec2, C.ptr(y1.g), None, True, y1, lane,
                             reduce_from=C.ptr(y2.g))
        y1.ginit = y1.gmasked = True
        # conv1: its input is the block input x; the residual stream joins before the mask
        self.bwd.tags[len(self.bwd)] = crec1.prefix
        self._fused_conv_bwd(C.ptr(y1.g), y1, xin1, crec1, C.ptr(x.g), residual, True,
                             self._rows_target(x, xin1), lane, reduce_from=C.ptr(y1.g))
        x.ginit = x.gmasked = True
        if lane == 0 and not in_region:
            self._bucket_mark_after_conv(crec1)

    def _fused_block_backward(self, ti, lane, in_region):
        _, terms, shifts, relu_out, out = self.tape[ti]
        _, xin2, crec2, _, y2, bn2 = self.tape[ti - 1]
        _, xin1, crec1, _, y1, bn1 = self.tape[ti - 2]
        x = xin1.act
        self.bwd.tags[len(self.bwd)] = crec2.prefix
        if not out.ginit:
            raise RuntimeError('no gradient reaches ' + out.name)
        assert not x.ginit and not y1.ginit and not y2.ginit
        if not out.gmasked:
            # the block output's gradient came from unfused consumers: apply its ReLU mask once, in place
            self.bwd.add(C.OP_GRAD_TERM, ints=(self.dtid, out.N, out.H, out.W, out.C, 0, 0, 0, 0),
                         ptrs=(C.ptr(out.g), C.ptr(out.g), C.ptr(out.t), None, None, None, None, None))
            out.gmasked = True
        # conv2: dz = d(out) masked (it is also the identity branch's gradient); its input is relu(bn1(y1))
        self._fused_conv_bwd(C.ptr(out.g), y2, xin2, crec2, C.ptr(y1.g), None, True, y1, lane,
                             reduce_from=C.ptr(out.g))
        y1.ginit = y1.gmasked = True
        y2.ginit = True
        # conv1: its input is the block input x; the residual stream (the masked d(out)) joins before the mask
        target = None
        if x is not self.inter_act:
            if xin1.bn is not None:
                target = x                       # x is a raw conv output read through its BatchNorm + ReLU
            else:
                ps = self._producer_sum.get(id(x))
                cand = [t for t, sh in zip(ps[1], ps[2])
                        if t.act.bn is not None and t.act.nuse == 1 and sh == 0 and not t.relu] if ps else []
                target = cand[0].act if cand else None
        self._fused_conv_bwd(C.ptr(y1.g), y1, xin1, crec1, C.ptr(x.g), C.ptr(out.g), True, target, lane)
        x.ginit = x.gmasked = True
        if lane == 0 and not in_region:
            self._bucket_mark_after_conv(crec1)

    def _emit_deferred_wgrads(self):
        """enqueue the recorded weight-gradient launches on the side lanes (largest first, least-loaded lane)"""
        # the deferred launches are small grids (32-64 workgroups each): more of them in flight than the three
        # branch lanes fill more of the chip beside the single-lane tail
        side = [l for l in range(1, 1 + self.defer_lanes)]
        if not self._deferred or not side:
            return
        keep = self.bwd.lane
        self.bwd.lane = 0
        self.bwd.fork(side)

        def emit(entries):
            load = {l: 0.0 for l in side}
            for wints, wptrs, ent, cost, _off in sorted(entries, key=lambda d: -d[3]):
                l = min(side, key=lambda q: load[q])
                load[l] += cost
                self.bwd.add(C.OP_WGRAD, ints=wints, ptrs=wptrs, lane=l)
                if ent is not None:
                    self._wred.setdefault(l, []).append(ent)
        if self.dp_plan and _knob('HRNET_LATE_GROUPS', '1') != '0':
            # Data parallelism: the late region of the flat gradient (net._flatten: module by module) leaves in GROUPS of
            # ~16 MB while the tail runs, instead of in one piece when the program ends. Group by group (highest
            # offsets first: the modules the backward pass finished first): the group's deferred launches spread over
            # the side lanes, (slab mode: their ordered sums,) then the other side lanes are joined into the first one,
            # so that "everything enqueued on side[0] so far" covers the group - and every late gradient that was NOT
            # deferred, which the fork above put in front. hipnet.optim.GradSync issues the group's all-reduce from
            # that stream at the recorded op index (late_cuts): the exchange of group g travels while the launches of
            # group g + 1 run, and nothing of the late region is left when the program ends.
            net = self.net
            lo_all, hi_all = int(net.late_start), int(net.trainable_count)
            starts = sorted({off for (off, n) in (net.offsets[id(p)] for p in net.params if net.is_late(p))})
            target = int(_knob('HRNET_LATE_GROUP_MB', '16')) << 18            # floats per group
            bounds = [hi_all]
            for off in reversed(starts):
                if bounds[-1] - off >= target:
                    bounds.append(off)
            if bounds[-1] != lo_all:
                if len(bounds) > 1 and bounds[-1] - lo_all < target // 2:
                    bounds[-1] = lo_all               # a small remainder joins the last group
                else:
                    bounds.append(lo_all)
            for hi_g, lo_g in zip(bounds, bounds[1:]):
                emit([d for d in self._deferred if lo_g <= d[4] < hi_g])
                for l in side:
                    self._flush_wred(l)
                for l in side[1:]:
                    self.bwd.sync(l, side[0])
                self.late_cuts.append((len(self.bwd), lo_g, hi_g, side[0]))
            self.bwd.lane = 0
        else:
            emit(self._deferred)
        self._deferred_lanes = side
        self.n_deferred_wgrads = len(self._deferred)
        self._deferred = []
        self.bwd.lane = keep

    def _bucket_mark_after_conv(self, crec):
        if self.batch_wred:
            if self._wred_bytes >= (16 << 20):
                for l in sorted(self._wred):
                    self._flush_wred(l)
                self.bwd.lane = 0
                if self.wlane:
                    self.bwd.sync(self.wlane, 0)
                self.bucket_marks.append((len(self.bwd), crec.prefix))
        else:
            if self.wlane:
                self.bwd.sync(self.wlane, 0)    # this bucket's weight gradients are complete
            self.bucket_marks.append((len(self.bwd), crec.prefix))

    def _flush_wred(self, lane):
        ents = self._wred.pop(lane, None)
        if not ents:
            return
        block = 0
        for e in ents:
            e['block0'] = block
            block += (e['Cout'] * e['Cin'] * e['ks'] * e['ks'] + 63) // 64
        keep = self.bwd.lane
        self.bwd.lane = lane
        i = self.bwd.add(C.OP_WGRAD_REDUCE_TABLE, ints=(len(ents), block), ptrs=(None,))
        self.bwd.lane = keep
        self._wred_tables.append((i, ents))
        if lane == 0 or lane == self.wlane:
            self._wred_bytes = 0

    def _upload_wred_tables(self):
        import ctypes
        n = sum(len(e) for _, e in self._wred_tables)
        if not n:
            return
        arr = (C.HrWredEnt * n)()
        k = 0
        starts = []
        for i, ents in self._wred_tables:
            starts.append((i, k))
            for e in ents:
                for name, val in e.items():
                    setattr(arr[k], name, val)
                k += 1
        raw = bytes(ctypes.string_at(ctypes.addressof(arr), ctypes.sizeof(arr)))
        table = torch.frombuffer(bytearray(raw), dtype=torch.uint8).to(self.dev)
        self.keep.append(table)
        for i, k0 in starts:
            self.bwd.ops[i].p[0] = C.ptr(table) + k0 * ctypes.sizeof(C.HrWredEnt)

    def _resolve_scratch(self):
        sizes = {'stats': max(self.max_stats, 1), 'slab': max(self.max_slab, 1), 'bwdpart': max(self.max_bwd_part, 1)}
        table = {}
        for prog, idx, slot, kind in self.pending:
            key = (kind, prog.lane_of(idx))    # concurrent lanes must not share scratch
            if key not in table:
                table[key] = self._f32(sizes[kind])
            prog.ops[idx].p[slot] = C.ptr(table[key])
        self.pending = []
        self.bwd.lane = 0

    def _side_streams(self):
        if self.nlanes <= 1:
            return None
        if self.streams is None:
            # lane 0 = PyTorch's current stream. (Measured on MI355X: high-priority side streams made the
            # backward 2.5x slower, and a separate weight-gradient lane 7 % slower - its big grids take
            # CU slots from the critical path - so both stay off by default.)
            self.streams = [None] + [torch.cuda.Stream(device=self.dev)
                                     for _ in range(max(self.nlanes, self.defer_lanes))]
        return self.streams

    # ---- execution --------------------------------------------------------------------------
    def run_forward(self, x):
        self.gen += 1
        N = self.N
        oa, ia = self.out_act, self.inter_act
        hm = torch.empty((N, self.nj, oa.H, oa.W), dtype=torch.float32, device=self.dev)
        inter = torch.empty((N, ia.C, ia.H, ia.W), dtype=torch.float32, device=self.dev)
        self.fwd.set_ptr(self.in_op, 0, x.data_ptr())
        self.fwd.set_ptr(self.out_op, 1, hm.data_ptr())
        self.fwd.set_ptr(self.inter_op, 1, inter.data_ptr())
        self.fwd.run(streams=self._side_streams())
        return hm, inter

    def run_backward(self, g_hm, g_inter=None, segment_hook=None):
        self.bwd.set_ptr(self.gout_op, 0, g_hm.data_ptr())
        if g_inter is not None:
            # d(inter_feat) joins the gradient of stage3's branch-0 output before its consumers'
            # contributions are read: run up to that op, add it, continue
            cut = self.inter_gop
            self._run_segments(0, cut, segment_hook, last=False)     # stage 4's buckets are exchanged on the way
            ia = self.inter_act
            tmp = torch.empty(ia.N * ia.H * ia.W * ia.C * self.esize, dtype=torch.uint8, device=self.dev)
            C.call('hrnet_nchw_to_nhwc', self.dtid, g_inter.data_ptr(), tmp.data_ptr(), ia.N, ia.H, ia.W, ia.C,
                   ia.C, C.stream_ptr())
            C.call('hrnet_grad_term', self.dtid, ia.g.data_ptr(), tmp.data_ptr(), None, None, None, None, None,
                   ia.N, ia.H, ia.W, ia.C, 0, 0, 1, C.stream_ptr())
            self._run_segments(cut, len(self.bwd), segment_hook)
        else:
            self._run_segments(0, len(self.bwd), segment_hook)

    def _run_segments(self, lo, hi, hook, last=True):
        streams = self._side_streams()
        if hook is None:
            self.bwd.run(lo, hi, streams=streams)
            return
        cuts = [c for c in hook.cuts if lo < c <= hi and (c < hi or not last)]
        prev = lo
        for c in cuts:
            self.bwd.run(prev, c, streams=streams)
            hook.after(c)
            prev = c
        self.bwd.run(prev, hi, streams=streams)
        if last:
            hook.after(hi)
