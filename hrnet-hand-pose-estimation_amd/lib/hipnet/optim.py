"""Optimizer + data-parallel gradient exchange over the flat parameter/gradient buffers.

FlatAdam   torch.optim.Adam semantics (lib/utils/utils.py:81-85: lr, L2 weight decay added to the
           gradient; betas (0.9, 0.999), eps 1e-8) as ONE fused HIP kernel over the flat f32 master
           parameters instead of 921 per-tensor updates.
GradSync   replaces DistributedDataParallel's reducer (tools/train.py:239-244): parameters and buffers
           are broadcast from rank 0 when it is installed (as DDP's constructor does), then every step
           a sum-all-reduce of the flat gradient over RCCL, issued bucket by bucket while the backward
           program is still running (gradients complete in reverse layer order = descending flat
           offsets). The 1/world_size average is folded into FlatAdam's gradient scale when that is the
           optimizer; with any other optimizer finish() scales the flat gradient itself.
"""
import torch

from . import _capi as C


class FlatAdam(object):
    def __init__(self, model, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0):
        self.model = model
        self.lr, self.betas, self.eps, self.weight_decay = lr, betas, eps, weight_decay
        self.step_count = 0
        self.grad_scale = 1.0
        self._state_for = None
        self.param_groups = [{'lr': lr, 'initial_lr': lr, 'weight_decay': weight_decay, 'betas': betas, 'eps': eps}]

    def _state(self):
        net = self.model.hip()
        if self._state_for is not net:
            self.exp_avg = torch.zeros_like(net.flat_p)
            self.exp_avg_sq = torch.zeros_like(net.flat_p)
            self._state_for = net
        return net

    def zero_grad(self, set_to_none=True):
        net = self._state()
        if set_to_none:
            for p in net.params:
                p.grad = None
        else:
            net.flat_g.zero_()

    def step(self):
        net = self._state()
        self.step_count += 1
        lr = self.param_groups[0]['lr']
        C.call('hrnet_adam_step', net.flat_p.data_ptr(), net.flat_g.data_ptr(), self.exp_avg.data_ptr(),
               self.exp_avg_sq.data_ptr(), net.trainable_count, lr, self.betas[0], self.betas[1], self.eps,
               self.weight_decay, self.step_count, self.grad_scale, C.stream_ptr())
        net.mark_weights_dirty()

    def state_dict(self):
        """the moments are flat tensors in the order of HipNet's flat buffer ([main | late | frozen], net._flatten);
        `layout` = [(parameter name, offset, numel)] says which slice belongs to which parameter, so that a checkpoint
        written under another flat order is re-mapped by name on load instead of landing on the wrong parameters"""
        net = self._state()      # the moment buffers are created lazily: a checkpoint before the first step has zeros
        return {'step': self.step_count, 'exp_avg': self.exp_avg, 'exp_avg_sq': self.exp_avg_sq,
                'param_groups': self.param_groups, 'layout': net.layout()}

    def load_state_dict(self, sd):
        net = self._state()
        cur = net.layout()
        saved = sd.get('layout')
        if saved is None:
            # a checkpoint written before the flat buffer was re-ordered: trainable parameters in module order
            saved = legacy_layout(cur, net.param_order())
        self.step_count = sd['step']
        self.exp_avg.copy_(remap_flat(sd['exp_avg'], saved, cur, self.exp_avg.numel()))
        self.exp_avg_sq.copy_(remap_flat(sd['exp_avg_sq'], saved, cur, self.exp_avg_sq.numel()))
        self.param_groups = sd['param_groups']


def legacy_layout(cur_layout, module_order):
    """[(name, offset, numel)] of the flat order used before the [main | late | frozen] permutation: the trainable
    parameters in module (named_parameters) order, frozen ones behind them (they carry no moments)"""
    size = {name: n for name, _o, n in cur_layout}
    out, off = [], 0
    for name in module_order:
        if name in size:
            out.append((name, off, size[name]))
            off += size[name]
    return out


def remap_flat(saved, saved_layout, cur_layout, cur_numel):
    """flat per-parameter state `saved` (laid out as saved_layout) re-ordered into cur_layout; refuses a checkpoint
    whose parameter names or sizes differ from this model's"""
    saved_layout = [(str(a), int(b), int(c)) for a, b, c in saved_layout]
    cur_layout = [(str(a), int(b), int(c)) for a, b, c in cur_layout]
    if saved_layout == cur_layout and saved.numel() == cur_numel:
        return saved
    smap = {name: (off, n) for name, off, n in saved_layout}
    if set(smap) != {name for name, _o, _n in cur_layout}:
        missing = sorted({name for name, _o, _n in cur_layout} ^ set(smap))[:5]
        raise ValueError('optimizer checkpoint does not match this model (parameter names differ: {} ...)'.format(missing))
    out = saved.new_zeros(cur_numel)
    for name, off, n in cur_layout:
        so, sn = smap[name]
        if sn != n or so + sn > saved.numel():
            raise ValueError('optimizer checkpoint does not match this model ({}: {} vs {} elements)'.format(name, sn, n))
        out[off:off + n] = saved[so:so + sn]
    return out


class GradSync(object):
    """Bucketed all-reduce of the flat gradient, overlapped with the backward program.

    Installed as `model._segment_hook`; engine.Plan._run_segments calls `after(op_index)` right
    after enqueueing a backward segment."""

    def __init__(self, model, optimizer=None, bucket_bytes=16 << 20, process_group=None, broadcast=True):
        import torch.distributed as dist
        self.dist = dist
        self.pg = process_group
        self.world = dist.get_world_size(process_group)
        self.model = model
        self.bucket_bytes = bucket_bytes
        self.cuts = []
        self._ranges = {}
        self._works = []
        self._plan = None
        self._done = set()
        self.scale_in_finish = True
        model._segment_hook = self
        if optimizer is not None:
            self.attach(optimizer)
        if broadcast:
            self.broadcast_state()

    def attach(self, optimizer):
        """fold the 1/world average into the optimizer when it can take it (FlatAdam.grad_scale); otherwise
        finish() divides the reduced gradient (torch.optim.SGD / AdamW / HRNET_TORCH_OPTIM=1)"""
        if hasattr(optimizer, 'grad_scale'):
            optimizer.grad_scale = 1.0 / self.world
            self.scale_in_finish = False
        else:
            self.scale_in_finish = True

    def broadcast_state(self, src=0):
        """every replica starts from rank `src`'s parameters and BatchNorm buffers (DDP's constructor semantics)"""
        net = self.model.hip()
        self.dist.broadcast(net.flat_p, src, group=self.pg)
        for b in self.model.buffers():
            self.dist.broadcast(b, src, group=self.pg)
        net.mark_weights_dirty()

    def _prepare(self, plan):
        net = plan.net
        total = net.total_params
        self.cuts, self._ranges, self._late, self._late_ranges = [], {}, None, {}
        deferred = getattr(plan, 'defer_wgrad', False) or getattr(plan, 'offload_wgrad', False)
        # recorded for data parallelism (engine.Plan.dp_plan): only gradients of the flat buffer's late region finish
        # out of order (deferred weight-gradient launches), nothing is offloaded from lane 0
        dp_ok = bool(getattr(plan, 'dp_plan', False)) and not getattr(plan, 'offload_wgrad', False)
        hi = total
        if deferred and not dp_ok:
            # this plan was recorded for a single process (before init_process_group, or while the default group had
            # one rank): its deferred / offloaded weight-gradient launches finish only when the program ends, so no
            # range is final at a mark - exchange everything once, after the last op
            import warnings
            warnings.warn('GradSync: the backward program was recorded with deferred weight gradients (single-process '
                          'plan); the gradient is exchanged in one all-reduce after the pass instead of in overlapped '
                          'buckets. Build the model after torch.distributed.init_process_group for the overlapped form.')
            marks = []
        else:
            marks = plan.bucket_marks
            if deferred:
                hi = int(net.late_start)          # the main region in buckets at the marks ...
                groups = list(getattr(plan, 'late_cuts', []) or [])
                covered = sorted((lo, hi_) for _i, lo, hi_, _l in groups)
                whole = (bool(covered) and covered[0][0] == hi and covered[-1][1] == int(net.trainable_count)
                         and all(a[1] == b[0] for a, b in zip(covered, covered[1:])))
                if whole:
                    # ... and the late region group by group while the single-lane tail of the pass runs
                    # (engine.Plan._emit_deferred_wgrads): each group from the side lane it is final on
                    for op_index, lo, hi_, lane in groups:
                        self._late_ranges[int(op_index)] = (int(lo), int(hi_), int(lane))
                    if total > int(net.trainable_count):
                        self._late = (int(net.trainable_count), total)      # (frozen parameters: zeros, one small range)
                else:
                    self._late = (hi, total)      # ... the late region in one piece when the program ends
        top = hi
        last = hi
        fallback = False
        # (with a late region the main region is a sixth of the gradient - 17 MB for w32: smaller buckets keep its
        # exchange overlapped with the backward pass instead of leaving it to the end with the late region)
        bucket_bytes = min(self.bucket_bytes, 4 << 20) if (self._late is not None or self._late_ranges) else self.bucket_bytes
        for op_index, prefix in marks:
            w = net.convs[prefix].mod.weight
            off = net.offsets[id(w)][0]
            if off > last:        # not monotone: fall back to one exchange at the end
                self.cuts, self._ranges = [], {}
                hi = top
                fallback = True
                break
            last = off
            if (hi - off) * 4 >= bucket_bytes:
                self.cuts.append(op_index)
                self._ranges[op_index] = (off, hi)
                hi = off
        if self._late_ranges:
            if fallback:
                self._late_ranges = {}            # (non-monotone marks: everything in one exchange at the end)
                self._late = (hi, total)
            else:
                self.cuts = sorted(set(self.cuts) | set(self._late_ranges))
        self._tail = (0, hi)
        self._end = len(plan.bwd)
        self._plan = plan

    def describe(self):
        """the exchange plan of the last backward program: op index of every cut and the f32 range it sends"""
        if self._plan is None:
            return None
        rows = [{'after_op': int(i), 'offset': int(lo), 'floats': int(hi - lo), 'mb': round((hi - lo) * 4 / 1e6, 2)}
                for i, (lo, hi) in sorted(self._ranges.items())]
        rows += [{'after_op': int(i), 'offset': int(lo), 'floats': int(hi - lo), 'mb': round((hi - lo) * 4 / 1e6, 2),
                  'late_group': True, 'lane': int(lane)} for i, (lo, hi, lane) in sorted(self._late_ranges.items())]
        rows.append({'after_op': int(self._end), 'offset': int(self._tail[0]),
                     'floats': int(self._tail[1] - self._tail[0]),
                     'mb': round((self._tail[1] - self._tail[0]) * 4 / 1e6, 2)})
        if self._late is not None and self._late[1] > self._late[0]:
            rows.append({'after_op': int(self._end), 'offset': int(self._late[0]),
                         'floats': int(self._late[1] - self._late[0]),
                         'mb': round((self._late[1] - self._late[0]) * 4 / 1e6, 2), 'late_region': True})
        exposed = sum(r['mb'] for r in rows if r['after_op'] == int(self._end))
        return {'payload': 'f32', 'bucket_bytes_min': int(self.bucket_bytes), 'backward_ops': int(self._end),
                'exposed_mb_after_backward': round(exposed, 2), 'buckets': rows}

    def begin(self, plan):
        if self._plan is not plan:
            self._prepare(plan)
        self._works = []
        self._done = set()

    def after(self, op_index):
        net = self._plan.net
        late = self._late_ranges.get(op_index)
        if late is not None and ('late', op_index) not in self._done:
            # a group of the late region: final on side lane `lane` at this point of the program (the other side lanes
            # were joined into it), so the collective is issued with THAT stream current - it waits for the group's
            # launches, not for lane 0's tail
            self._done.add(('late', op_index))
            lo, hi, lane = late
            streams = getattr(self._plan, 'streams', None)
            st = streams[lane] if streams is not None and lane < len(streams) and streams[lane] is not None else None
            if st is not None:
                with torch.cuda.stream(st):
                    self._works.append(self.dist.all_reduce(net.flat_g[lo:hi], group=self.pg, async_op=True))
            else:
                self._works.append(self.dist.all_reduce(net.flat_g[lo:hi], group=self.pg, async_op=True))
        rng = self._ranges.get(op_index)
        if rng is None and op_index == self._end:
            rng = self._tail
        if rng is None or op_index in self._done:
            return
        self._done.add(op_index)
        if rng[1] > rng[0]:
            self._works.append(self.dist.all_reduce(net.flat_g[rng[0]:rng[1]], group=self.pg, async_op=True))
        if op_index == self._end and self._late is not None and self._late[1] > self._late[0]:
            # (the late region: its deferred weight-gradient launches were joined into this stream by the program's
            # last ops; in two halves so that the first travels while the second is being queued)
            lo, hi = self._late
            mid = lo + (hi - lo) // 2
            for a, b in ((lo, mid), (mid, hi)):
                if b > a:
                    self._works.append(self.dist.all_reduce(net.flat_g[a:b], group=self.pg, async_op=True))

    def finish(self):
        """wait for the exchanges of this step; any bucket whose mark the backward run did not pass (it was
        executed in pieces around an external gradient) is exchanged now, so no range stays rank-local"""
        if self._plan is not None:
            for op_index in sorted(set(self._ranges) | set(self._late_ranges)) + [self._end]:
                self.after(op_index)              # (idempotent: ranges already exchanged are skipped)
        for w in self._works:
            w.wait()
        self._works = []
        if self.scale_in_finish and self._plan is not None and self.world > 1:
            self._plan.net.flat_g.mul_(1.0 / self.world)
