"""Device-side state of one PoseHighResolutionNet: flat f32 master parameters / gradients,
packed (f32 or bf16) kernel weights, BatchNorm coefficient buffers and the cached plans."""
import torch
import torch.nn as nn

from . import _capi as C
from .engine import BNRec, ConvRec, Plan, Program


class HipNet(object):
    def __init__(self, module, stage_cfg, compute_dtype):
        self.module = module
        self.stage_cfg = stage_cfg          # {2: STAGE2 cfg, 3: ..., 4: ...}
        self.compute_dtype = compute_dtype
        self.dtid = C.dtype_id(compute_dtype)
        C.lib()                              # fail loudly now if the HIP library is missing
        params = list(module.parameters())
        if not params or not params[0].is_cuda:
            raise RuntimeError('pose_hrnet: parameters must live on a HIP device (model.cuda()); '
                               'there is no CPU path')
        self.device = params[0].device
        self.plans = {}
        self.fwd_packed = self.bwd_packed = False
        self._flatten(params)
        self._records()

    # ---- flat master parameters / gradients --------------------------------------------------
    def _flatten(self, params):
        # frozen parameters (requires_grad=False, e.g. a fixed softmax temperature) sit behind the trainable
        # ones so that the fused optimiser kernel covers exactly [0, trainable_count)
        # ... and, among the trainable ones, the convolution weights whose gradients the backward program may finish
        # LATE (the deferred weight-gradient launches of the HighResolutionModules: wide branches and fuse layers) sit
        # behind all the others: [main region, forward order | late region | frozen]. A data-parallel run exchanges
        # the main region in buckets overlapped with the backward pass (every gradient above a flat offset is final at
        # a bucket mark) and the late region when the program ends (hipnet.optim.GradSync).
        names = {id(p): n for n, p in self.module.named_parameters()}
        self._late_ids = {id(p) for p in params if p.requires_grad and self._is_late(names.get(id(p), ''), p)}
        params = ([p for p in params if p.requires_grad and id(p) not in self._late_ids]
                  + [p for p in params if id(p) in self._late_ids] + [p for p in params if not p.requires_grad])
        self.trainable_count = sum(p.numel() for p in params if p.requires_grad)
        self.late_start = sum(p.numel() for p in params if p.requires_grad and id(p) not in self._late_ids)
        total = sum(p.numel() for p in params)
        self.flat_p = torch.empty(total, dtype=torch.float32, device=self.device)
        self.flat_g = torch.zeros(total, dtype=torch.float32, device=self.device)
        self.params = params
        self._gview = {}
        self.offsets = {}
        off = 0
        with torch.no_grad():
            for p in params:
                n = p.numel()
                self.flat_p[off:off + n].copy_(p.detach().reshape(-1).float())
                p.data = self.flat_p[off:off + n].view(p.shape)
                self._gview[id(p)] = self.flat_g[off:off + n].view(p.shape)
                self.offsets[id(p)] = (off, n)
                if p.grad is not None:
                    self._gview[id(p)].copy_(p.grad)
                    p.grad = self._gview[id(p)]
                off += n
        self.total_params = total

    def grad_of(self, p):
        return self._gview[id(p)]

    def layout(self):
        """[(parameter name, flat offset, numel)] of the trainable parameters, in flat order (FlatAdam tags its
        checkpoints with it)"""
        names = {id(p): n for n, p in self.module.named_parameters()}
        return [(names[id(p)],) + self.offsets[id(p)] for p in self.params if p.requires_grad]

    def param_order(self):
        """parameter names in module order (the flat order of checkpoints written before the late region existed)"""
        return [n for n, _p in self.module.named_parameters()]

    @staticmethod
    def _is_late(name, p):
        """conv weights of a HighResolutionModule's wide branches (index >= 2) and fuse layers (see _flatten)"""
        if p.dim() != 4 or not name.startswith('stage') or not name.endswith('.weight'):
            return False
        parts = name.split('.')
        if 'fuse_layers' in parts:
            return True
        return 'branches' in parts and int(parts[parts.index('branches') + 1]) >= 2

    def is_late(self, p):
        return id(p) in self._late_ids

    def _records(self):
        self.convs, self.bns, self.bn_list, self.bias_pad = {}, {}, {}, {}
        self.bn_list = []
        for name, m in self.module.named_modules():
            if isinstance(m, nn.Conv2d):
                r = ConvRec(name, m, stem=(name == 'conv1'))
                es = torch.empty((), dtype=self.compute_dtype).element_size()
                taps = r.ks * r.ks
                nf = r.Cout_pad * (1 if r.stem else taps) * r.Cin_pad
                r.wf = torch.empty(nf * es, dtype=torch.uint8, device=self.device)
                if not r.stem:
                    r.wd = torch.empty(r.Cin_pad * taps * r.Cout_pad * es, dtype=torch.uint8, device=self.device)
                self.convs[name] = r
                if m.bias is not None:
                    self.bias_pad[name] = torch.zeros(r.Cout_pad, dtype=torch.float32, device=self.device)
            elif isinstance(m, nn.BatchNorm2d):
                b = BNRec(name, m, self.device)
                self.bns[name] = b
                self.bn_list.append(b)
        # one launch packs every convolution (forward layouts), one more the transposed dgrad copies
        self.pack_f = self._pack_program([(r, 2 if r.stem else 0, r.wf) for r in self.convs.values()])
        self.pack_d = self._pack_program([(r, 1, r.wd) for r in self.convs.values() if r.wd is not None])

    def _pack_program(self, items):
        import ctypes
        ents = (C.HrPackEnt * len(items))()
        block = 0
        for e, item in zip(ents, items):
            r, mode, out = item[:3]
            taps = r.ks * r.ks
            if (mode == 1 and r.Cout_pad * (taps + 1) > 4864) or (mode == 0 and r.Cin * taps > 4864):
                raise ValueError('{}: {} output channels at {}x{} exceed the LDS staging of the weight packer'.format(
                    r.prefix, r.Cout, r.ks, r.ks))
            e.w, e.out = C.ptr(r.mod.weight), C.ptr(out)
            e.Cout, e.Cin, e.ks, e.Cout_pad, e.Cin_pad, e.mode = r.Cout, r.Cin, r.ks, r.Cout_pad, r.Cin_pad, mode
            if len(item) > 3:
                # a column slice [c0, c0 + c) of a 1x1 weight: rows r.Cin floats apart
                c0, c = item[3], item[4]
                assert r.ks == 1 and mode == 0
                e.w = r.mod.weight.data_ptr() + 4 * c0
                e.Cin, e.Cin_pad, e.ld = c, c, r.Cin
            e.block0 = block
            block += C.call('hrnet_pack_blocks', e.Cout_pad, e.Cin_pad, r.ks, mode)
        raw = bytes(ctypes.string_at(ctypes.addressof(ents), ctypes.sizeof(ents)))
        table = torch.frombuffer(bytearray(raw), dtype=torch.uint8).to(self.device)
        self._tables = getattr(self, '_tables', []) + [table]
        prog = Program()
        prog.add(C.OP_PACK_TABLE, ints=(self.dtid, len(items), block), ptrs=(C.ptr(table),))
        return prog.finalize()

    def head_slices(self, crec, widths):
        """forward-layout copies [Cout][C_j] of the column slices W[:, c_j : c_j + C_j] of a 1x1 weight (the head
        without its concat, engine.Plan.head_mix): packed by one more table launch whenever the weights are"""
        key = (crec.prefix, tuple(widths))
        if getattr(self, '_head_key', None) != key:
            es = torch.empty((), dtype=self.compute_dtype).element_size()
            items, off = [], 0
            self._head_wf = []
            for c in widths:
                out = torch.empty(crec.Cout_pad * c * es, dtype=torch.uint8, device=self.device)
                self._head_wf.append(out)
                items.append((crec, 0, out, off, c))
                off += c
            self._head_pack = self._pack_program(items)
            self._head_key = key
            self._head_pack.run()
        return self._head_wf

    def pack_weights(self, for_backward):
        """master f32 OIHW -> kernel layouts (forward always, transposed dgrad copy on demand)"""
        if not self.fwd_packed:
            self.pack_f.run()
            if getattr(self, '_head_pack', None) is not None:
                self._head_pack.run()
            for name, buf in self.bias_pad.items():
                b = self.convs[name].mod.bias
                buf[:b.numel()].copy_(b.detach())
            self.fwd_packed = True
        if for_backward and not self.bwd_packed:
            self.pack_d.run()
            self.bwd_packed = True

    def mark_weights_dirty(self):
        self.fwd_packed = self.bwd_packed = False

    # ---- forward / backward -------------------------------------------------------------------
    MAX_PLANS_IN_FLIGHT = 4

    def plan(self, N, H, W, training, need_grad):
        """the cached plan of this shape. A plan owns its activation / gradient buffers, so a training forward
        whose backward has not run yet keeps it busy: a second forward of the same shape in between (gradient
        accumulation over summed losses, siamese / consistency losses) gets its own plan instead of overwriting
        the activations the first graph's backward will read."""
        key = (N, H, W, bool(training), bool(need_grad))
        pool = self.plans.setdefault(key, [])
        for p in pool:
            if not (need_grad and p.busy):
                return p
        if len(pool) >= self.MAX_PLANS_IN_FLIGHT:
            raise RuntimeError('pose_hrnet: {} training forwards of shape {} are waiting for their backward; run '
                               'backward (or drop the outputs) before more forwards'.format(len(pool), (N, H, W)))
        p = Plan(self, N, H, W, training, need_grad)
        pool.append(p)
        return p

    def forward(self, x, training, need_grad):
        if x.dim() != 4 or x.shape[1] != 3:
            raise ValueError('pose_hrnet expects (B,3,H,W) input, got {}'.format(tuple(x.shape)))
        N, _, H, W = x.shape
        if H % 32 or W % 32:
            raise ValueError('pose_hrnet: input height/width must be multiples of 32, got {}x{}'.format(H, W))
        if need_grad and not training:
            raise RuntimeError('pose_hrnet: backward through eval-mode BatchNorm is not implemented')
        self.pack_weights(for_backward=need_grad)
        plan = self.plan(N, H, W, training, need_grad)
        hm, inter = plan.run_forward(x)
        return hm, inter, plan

    def all_plans(self):
        return [p for pool in self.plans.values() for p in pool]

    def prepare_grads(self):
        """PyTorch semantics for .grad: None -> fresh, ours -> accumulate, a foreign tensor (autograd
        reached a parameter outside the recorded programs first, e.g. the softmax temperature) ->
        adopted into the flat buffer."""
        ours = [p.grad is not None and p.grad.data_ptr() == self._gview[id(p)].data_ptr() for p in self.params]
        if not any(ours):
            foreign = [(p, p.grad) for p in self.params if p.grad is not None]
            self.flat_g.zero_()
            for p, g in foreign:
                self._gview[id(p)].copy_(g)
            for p in self.params:
                p.grad = self._gview[id(p)]
            return
        for p, mine in zip(self.params, ours):
            if p.grad is None:
                self._gview[id(p)].zero_()
                p.grad = self._gview[id(p)]
            elif not mine:
                self._gview[id(p)].copy_(p.grad)
                p.grad = self._gview[id(p)]

    def backward(self, plan, g_hm, g_inter, segment_hook=None):
        self.prepare_grads()
        if segment_hook is not None:
            segment_hook.begin(plan)
        plan.run_backward(g_hm, g_inter, segment_hook)
        self.mark_weights_dirty()   # an optimizer step normally follows
