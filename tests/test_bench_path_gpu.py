"""End-to-end GPU parity on the code path behind the headline number, and bf16 fidelity tests that can fail.

* B=40 256x256 (every 64x64 / 32x32 conv walks 2..5 pixel tiles per workgroup, as at the benchmark's B=64)
  training step, fp32 device path, against the CPU oracle (reference: lib/models/pose_hrnet.py:511-568,
  lib/core/loss.py:19-28): heat maps <= 1e-3 max-abs, loss 2e-5 relative, gradients in the fp32 band.
* the same step in bf16 (the dtype the headline is quoted in) against the fp32 device path.
* bf16 fidelity on the reference's own init_weights (pose_hrnet.py:570-600: conv N(0, 0.001), BatchNorm
  gamma 1 / beta 0, fresh running statistics): first-step gradient cosine against the fp64 oracle beside
  PyTorch's CPU bf16 kernels, a 30-step loss trajectory of bf16 against fp32 on the synthetic loader, and -
  because a randomly initialised 70-layer BatchNorm stack is chaotic end to end - every op of the recorded
  programs in bf16 on exact inputs against the fp32 device path (the test that can fail).
* config 4 (w48) once in bf16.
"""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXP = os.path.join(REPO, 'hrnet-hand-pose-estimation_amd', 'experiments', 'RHD')
YAML = os.path.join(EXP, 'RHD_HRNet_w32_max_hmloss_v1.yaml')
YAML48 = os.path.join(EXP, 'RHD_HRNet_w48_softmax_hm-pose2dloss_v1.yaml')


def _model(dtype, sd=None, salt=0, yaml=YAML, init='fill'):
    from config import get_cfg_defaults
    from hipnet import synth
    from models import pose_hrnet
    cfg = get_cfg_defaults()
    cfg.merge_from_file(yaml)
    cfg.MODEL.COMPUTE_DTYPE = dtype
    model = pose_hrnet.get_pose_net(cfg, is_train=False)
    if sd is None:
        if init == 'fill':
            sd = {k: torch.from_numpy(np.asarray(v)) for k, v in synth.fill_state_dict(model.state_dict(), salt).items()}
        else:
            # the reference's init_weights (pose_hrnet.py:570-600), seeded
            g = torch.Generator().manual_seed(1234 + salt)
            sd = {}
            for k, v in model.state_dict().items():
                leaf = k.rsplit('.', 1)[-1]
                if v.dim() == 4:
                    sd[k] = torch.randn(v.shape, generator=g) * 0.001
                elif leaf == 'num_batches_tracked':
                    sd[k] = torch.zeros((), dtype=torch.int64)
                elif leaf == 'running_var':
                    sd[k] = torch.ones_like(v)
                elif leaf == 'weight':
                    sd[k] = torch.ones_like(v)          # BatchNorm gamma
                else:
                    sd[k] = torch.zeros_like(v)         # biases, BatchNorm beta, running_mean
    model.load_state_dict({k: v.clone() for k, v in sd.items()}, strict=True)
    return model.cuda(), sd


def _oracle_step(sd, extra, batch, dtype, backward=True):
    from oracle import hrnet_cpu as O
    osd = {k: (v.clone().to(dtype) if v.is_floating_point() else v.clone()) for k, v in sd.items()}
    pkeys = [k for k in osd if not k.endswith(('running_mean', 'running_var', 'num_batches_tracked'))]
    if backward:
        for k in pkeys:
            osd[k].requires_grad_(True)
    x = torch.from_numpy(batch['imgs']).to(dtype)
    gt = torch.from_numpy(batch['heatmaps']).to(dtype)
    with torch.set_grad_enabled(backward):
        hm, inter, new_stats = O.hrnet_forward(osd, extra, x, training=True)
        loss = O.heatmap_loss(hm, gt)
    grads = None
    if backward:
        loss.backward()
        grads = {k: osd[k].grad.double() for k in pkeys}
    return dict(hm=hm.detach(), inter=inter.detach(), loss=float(loss.item()), stats=new_stats, grads=grads)


def _hip_step(model, batch):
    from core.loss import HeatmapLoss
    model.train()
    model.zero_grad(set_to_none=True)
    hm, inter = model(torch.from_numpy(batch['imgs']).cuda())
    loss = HeatmapLoss()(hm, torch.from_numpy(batch['heatmaps']).cuda())
    loss.backward()
    grads = {k: p.grad.detach().double().cpu() for k, p in model.named_parameters()}
    return hm.detach().cpu(), inter.detach().cpu(), float(loss.item()), grads


def _cos(ga, gb, keys=None):
    keys = keys or list(gb)
    dot = sum(float((ga[k] * gb[k]).sum()) for k in keys)
    na = np.sqrt(sum(float((ga[k] ** 2).sum()) for k in keys))
    nb = np.sqrt(sum(float((gb[k] ** 2).sum()) for k in keys))
    return dot / max(na * nb, 1e-300)


def _per_tensor_err(ga, gb):
    out = []
    top = max(v.abs().max().item() for v in gb.values())       # (once: inside the loop this was 921 x 921 reductions)
    for k, ref in gb.items():
        sc = ref.abs().max().item()
        if sc < 1e-6 * top:
            continue
        out.append((ga[k] - ref).abs().max().item() / sc)
    return np.array(out)


@pytest.fixture(scope='module')
def b40():
    """B=40 256x256 batch + the fp32 oracle's training step on it (CPU, ~20 s)"""
    from hipnet import synth
    from oracle import hrnet_cpu as O
    _, sd = _model('fp32', salt=0)
    batch = synth.rhd_batch(40, seed=4321)
    ref = _oracle_step(sd, O.W32_EXTRA, batch, torch.float32)
    return sd, batch, ref


def test_fp32_training_step_b40_256_on_the_multi_tile_walk(b40):
    from hipnet import _capi as C
    sd, batch, ref = b40
    model, _ = _model('fp32', sd)
    hm, inter, loss, grads = _hip_step(model, batch)
    # the recorded programs of this step really take the walk (tiles-per-workgroup >= 2) on most conv launches
    plan = model.hip().plan(40, 256, 256, True, True)
    import ctypes
    out = (ctypes.c_int * 5)()
    walked = total = 0
    for prog in (plan.fwd, plan.bwd):
        for op in prog.ops:
            if int(op.kind) == C.OP_CONV_SUM:      # conv with the residual sum in its prologue: same tile walk
                C.call('hrnet_conv_tile_walk', op.i[1], op.i[2], op.i[3], op.i[5], op.i[6], 1, 0, 0, out)
                total += 1
                walked += out[3] >= 2
            if int(op.kind) == C.OP_CONV:
                s2d = 1 if (op.i[10] and not op.p[2] and not op.p[4]) else 0
                C.call('hrnet_conv_tile_walk', op.i[1], op.i[5], op.i[6], op.i[7], op.i[8], op.i[9],
                       1 if op.p[7] else 0, s2d, out)
                total += 1
                walked += out[3] >= 2
    assert walked >= 150, (walked, total)        # every conv on the 64x64 and 32x32 maps (the small maps fit one tile each)
    assert (hm - ref['hm']).abs().max().item() <= 1e-3
    assert (inter - ref['inter']).abs().max().item() <= 1e-3
    assert abs(loss - ref['loss']) <= 2e-5 * abs(ref['loss'])
    # gradients: both sides are fp32 with different summation orders -> the ReLU-mask chaos band of DESIGN 2
    cos = _cos(grads, ref['grads'])
    errs = _per_tensor_err(grads, ref['grads'])
    print('B=40 fp32: grad cosine {:.6f}, per-tensor rel err median {:.2e} p95 {:.2e} max {:.2e}'.format(
        cos, np.median(errs), np.percentile(errs, 95), errs.max()))
    # measured 0.99985 (fp32 oracle vs fp32 device path, different summation orders); the batch sums are float
    # atomics in this mode, so the value moves from run to run (B=4 spread: 0.99975..0.99995)
    assert cos >= 0.999, cos
    assert np.median(errs) <= 2e-2 and np.percentile(errs, 95) <= 0.1, (np.median(errs), np.percentile(errs, 95))
    msd = model.state_dict()
    for k, v in ref['stats'].items():
        np.testing.assert_allclose(msd[k].cpu().numpy(), v.numpy(), rtol=1e-3, atol=1e-4, err_msg=k)


def test_bf16_training_step_b40_256_tracks_fp32(b40):
    """the headline dtype at a batch that takes the walk: bf16 device path against the fp32 oracle.
    Synthetic He-uniform weights with random BatchNorm affine are a hard case for 8-bit mantissas;
    the band below is what the measurement on this network gives, with margin, and fails on a wrong tile."""
    sd, batch, ref = b40
    model, _ = _model('bf16', sd)
    hm, inter, loss, grads = _hip_step(model, batch)
    rel = ((hm.double() - ref['hm'].double()).norm() / ref['hm'].double().norm()).item()
    cos = _cos(grads, ref['grads'])
    print('B=40 bf16: heat-map rel L2 {:.4f}, loss {:.4f} vs {:.4f}, grad cosine {:.4f}'.format(
        rel, loss, ref['loss'], cos))
    # end to end this random network is chaotic (see test_bf16_op_by_op_with_exact_inputs...): the whole-network
    # band is the one PyTorch's own CPU bf16 kernels reach on it (heat maps ~0.3 relative L2, gradient cosine
    # ~0.3); the tight bf16 statement is the op-by-op test below. Measured here: 0.29 / 0.36.
    assert rel <= 0.40, rel
    assert abs(loss - ref['loss']) <= 5e-3 * abs(ref['loss'])
    assert cos >= 0.25, cos


def test_bf16_first_step_gradient_cosine_on_reference_init():
    """(a) of the fidelity test: reference init_weights, B=8 128x128; bf16 device gradients against the
    fp64 oracle. PyTorch's own CPU bf16 autograd on the same network is printed beside it."""
    from hipnet import synth
    from oracle import hrnet_cpu as O
    model, sd = _model('bf16', init='reference', salt=1)
    batch = synth.rhd_batch(8, seed=77, img_h=128, img_w=128)
    r64 = _oracle_step(sd, O.W32_EXTRA, batch, torch.float64)
    hm, inter, loss, grads = _hip_step(model, batch)
    cos = _cos(grads, r64['grads'])
    m32, _ = _model('fp32', sd)
    _, _, loss32, g32 = _hip_step(m32, batch)
    cos32 = _cos(g32, r64['grads'])
    rb = _oracle_step(sd, O.W32_EXTRA, batch, torch.bfloat16)
    cos_t = _cos(rb['grads'], r64['grads'])
    rel = ((hm.double() - r64['hm']).norm() / r64['hm'].norm()).item()
    print('reference init: grad cosine vs fp64: hip-bf16 {:.5f}  hip-fp32 {:.6f}  torch-cpu-bf16 {:.5f}; '
          'heat-map rel L2 {:.4f}; loss {:.5f} / {:.5f}'.format(cos, cos32, cos_t, rel, loss, r64['loss']))
    # measured: hip-fp32 0.99982 (1e-7 roundings amplified to 2e-2 by ~70 BatchNorm layers at random init),
    # hip-bf16 0.339, torch-cpu-bf16 0.281: the device path must not be worse than PyTorch's own bf16 kernels
    assert cos32 >= 0.999, cos32
    assert cos >= cos_t - 0.05 and cos >= 0.2, (cos, cos_t)
    assert abs(loss - r64['loss']) <= 5e-3 * abs(r64['loss'])


def test_bf16_loss_trajectory_follows_fp32():
    """(b): 30 optimiser steps (Adam lr 1e-3, wd 1e-4, the yaml's values) on the synthetic loader from the
    reference init: the bf16 loss curve stays within a band of the fp32 curve, and both go down."""
    from core.loss import HeatmapLoss
    from hipnet import synth
    from hipnet.optim import FlatAdam
    curves = {}
    for dt in ('fp32', 'bf16'):
        model, _ = _model(dt, init='reference', salt=2)
        model.train()
        opt = FlatAdam(model, lr=1e-3, weight_decay=1e-4)
        crit = HeatmapLoss()
        losses = []
        for it in range(30):
            b = synth.rhd_batch(8, seed=500 + it % 6, img_h=128, img_w=128)     # 6 batches, cycled
            x, gt = torch.from_numpy(b['imgs']).cuda(), torch.from_numpy(b['heatmaps']).cuda()
            opt.zero_grad()
            hm, _ = model(x)
            loss = crit(hm, gt)
            loss.backward()
            opt.step()
            losses.append(float(loss.item()))
        curves[dt] = np.array(losses)
    f, b = curves['fp32'], curves['bf16']
    dev = np.abs(b - f) / f
    print('loss trajectory fp32 {} ... {}; bf16 {} ... {}; max rel deviation {:.4f}, mean {:.4f}'.format(
        np.round(f[:3], 3), np.round(f[-3:], 3), np.round(b[:3], 3), np.round(b[-3:], 3), dev.max(), dev.mean()))
    assert f[-6:].mean() < 0.9 * f[:6].mean()          # training makes progress at all
    assert b[-6:].mean() < 0.9 * b[:6].mean()
    # measured: max 0.24 (the two paths take the step-2 loss spike differently), mean 0.044, final 0.7 %
    assert dev.max() <= 0.40 and dev.mean() <= 0.08, (dev.max(), dev.mean())
    assert abs(b[-6:].mean() - f[-6:].mean()) <= 0.03 * f[-6:].mean()


def test_w48_bf16_training_step_tracks_fp32_device_path():
    """config 4 (w48: channels 48/96/192/384, non-square maps) once in bf16: against the fp32 device path
    (itself held to the oracle by test_model_gpu.py) on 192x160 crops, B=8."""
    from hipnet import synth
    m32, sd = _model('fp32', salt=8, yaml=YAML48, init='reference')
    batch = synth.rhd_batch(8, seed=17, img_h=192, img_w=160)
    hm32, _, loss32, g32 = _hip_step(m32, batch)
    m16, _ = _model('bf16', sd, yaml=YAML48)
    hm16, _, loss16, g16 = _hip_step(m16, batch)
    assert hm16.shape == (8, 21, 48, 40)
    rel = ((hm16.double() - hm32.double()).norm() / hm32.double().norm()).item()
    cos = _cos(g16, g32)
    print('w48 bf16 vs fp32 device path: heat-map rel L2 {:.4f}, loss {:.5f} / {:.5f}, grad cosine {:.5f}'.format(
        rel, loss16, loss32, cos))
    # whole-network band of a chaotic random-init stack (measured 0.33 / 0.29), as for w32 above
    assert rel <= 0.45, rel
    assert abs(loss16 - loss32) <= 5e-3 * abs(loss32)
    assert cos >= 0.2, cos


def test_config2_fp32_eval_b64_slice_matches_the_oracle():
    """BASELINE config 2 at its FULL size: pose_hrnet_w32 256x256 fp32 forward-only inference at batch 64
    on the device, arg-max decode; images 0-3 and the last image of that run against the CPU oracle's eval forward
    (reference lib/models/pose_hrnet.py:511-568 with running statistics) - heat maps / inter_feat <= 1e-3 max-abs
    (the north star's tolerance), decoded key points identical wherever the oracle's top two values differ by more
    than the tolerance. An eval forward is per-image independent, so a slice of the B=64 run IS the B=5 run."""
    from hipnet import synth
    from oracle import hrnet_cpu as O
    from utils.heatmap_decoding import get_final_preds
    # the reference fixture's calibrated BatchNorm statistics (tests/golden/make_golden.py: running stats := batch stats
    # of a calibration batch on the salt-0 weights), so that eval activations are O(1) and 1e-3 is a max-abs bound
    g = np.load(os.path.join(REPO, 'tests', 'golden', 'w32_eval_b1.npz'))
    _, sd = _model('fp32', salt=0)
    for k in g.files:
        if k.startswith('stat.'):
            sd[k[5:]] = torch.from_numpy(g[k])
    model, sd = _model('fp32', sd)
    model.eval()
    batch = synth.rhd_batch(64, seed=99)
    x = torch.from_numpy(batch['imgs'])
    with torch.no_grad():
        hm, inter = model(x.cuda())
        kp = get_final_preds(hm, use_softmax=False).cpu()
    assert hm.shape == (64, 21, 64, 64) and inter.shape == (64, 32, 64, 64)
    pick = [0, 1, 2, 3, 63]
    with torch.no_grad():
        ref_hm, ref_inter, _ = O.hrnet_forward({k: v.clone() for k, v in sd.items()}, O.W32_EXTRA, x[pick], training=False)
    err = (hm.cpu()[pick] - ref_hm).abs().max().item()
    ierr = (inter.cpu()[pick] - ref_inter).abs().max().item()
    print('config 2 (B=64 fp32 eval) vs oracle on images {}: heat-map max-abs {:.2e}, inter_feat {:.2e}'.format(pick, err, ierr))
    assert err <= 1e-3 and ierr <= 1e-3, (err, ierr)
    assert ref_hm.abs().max().item() < 1e3        # (calibrated: the bound above is a real max-abs bound)
    ref_kp = O.get_final_preds(ref_hm, use_softmax=False)
    top2 = ref_hm.reshape(len(pick), 21, -1).topk(2, dim=2).values
    clear = (top2[..., 0] - top2[..., 1]) > 2e-3
    assert clear.float().mean().item() > 0.5
    assert torch.equal(kp[pick][clear], ref_kp[clear])
    # nothing of the batch is left unwritten or shared between images
    assert torch.isfinite(hm).all() and (hm[4:63].flatten(1).std(dim=1) > 0).all()


def test_w48_config4_full_size_bf16_step_tracks_fp32_device_path():
    """BASELINE config 4 at its FULL size: pose_hrnet_w48 384x288, batch 32, one bf16 training step (forward,
    HeatmapLoss, backward) against the fp32 device path on the same batch and weights (the fp32 path is held to the
    oracle in test_model_gpu.py / above): loss within 1e-2 relative, heat maps and gradients inside the whole-network
    bf16 band measured on the 192x160 case, every gradient finite and non-zero (96x72 maps: tiles overhang on both
    axes, 48 / 96 / 192 / 384-channel instantiations, the K = 48 head)."""
    from hipnet import synth
    m32, sd = _model('fp32', salt=8, yaml=YAML48, init='reference')
    batch = synth.rhd_batch(32, seed=23, img_h=384, img_w=288)
    hm32, _, loss32, g32 = _hip_step(m32, batch)
    del m32
    torch.cuda.empty_cache()
    m16, _ = _model('bf16', sd, yaml=YAML48)
    hm16, _, loss16, g16 = _hip_step(m16, batch)
    assert hm16.shape == (32, 21, 96, 72)
    rel = ((hm16.double() - hm32.double()).norm() / hm32.double().norm()).item()
    cos = _cos(g16, g32)
    print('w48 384x288 B=32 bf16 vs fp32 device path: heat-map rel L2 {:.4f}, loss {:.5f} / {:.5f}, grad cosine {:.5f}'.format(
        rel, loss16, loss32, cos))
    assert abs(loss16 - loss32) <= 1e-2 * abs(loss32)
    assert rel <= 0.45, rel
    assert cos >= 0.2, cos
    for k, g in g16.items():
        assert torch.isfinite(g).all(), k
    dead = [k for k, g in g16.items() if g.abs().max().item() == 0.0 and g32[k].abs().max().item() > 0.0]
    assert not dead, dead[:5]
    # per-image heat maps: no image of the batch falls out of the band (a broken tile walk hits single images)
    per_img = ((hm16.double() - hm32.double()).flatten(1).norm(dim=1) / hm32.double().flatten(1).norm(dim=1))
    assert per_img.max().item() <= 0.6, per_img.max().item()


# ---------------------------------------------------------------------------------------------------------
# bf16 fidelity, op by op, with exact inputs ("teacher forcing")
# ---------------------------------------------------------------------------------------------------------
def _ptr_maps(plan, esize_dtype):
    acts = {}
    for i, a in enumerate(plan.acts):
        acts[a.t.data_ptr()] = ('t', i)
        if a.g is not None:
            acts[a.g.data_ptr()] = ('g', i)
    keep = {t.data_ptr(): t for t in plan.keep if t.dtype == torch.float32}
    for b in plan.bns.values():
        for t in (b.scale, b.shift, b.mean, b.invstd, b.coef):
            keep[t.data_ptr()] = t
    return acts, keep


def _act_view(plan, kind, i, dtype):
    a = plan.acts[i]
    return (a.t if kind == 't' else a.g).view(dtype)


def _table_jobs(op, plan, C):
    """the HrOp jobs of a table-driven launch (HR_OP_EW_TABLE): the device table is one of the plan's kept tensors"""
    import ctypes
    t = next(t for t in plan.keep if t.dtype == torch.uint8 and t.data_ptr() == op.p[0])
    raw = bytes(t.cpu().numpy().tobytes())
    n = int(op.i[0])
    return list((C.HrOp * n).from_buffer_copy(raw[:n * ctypes.sizeof(C.HrOp)]))


def _written(op, C, plan=None):
    """(pointer, is_activation) slots an op writes that later ops of the same program read"""
    k = int(op.kind)
    if k == C.OP_EW_TABLE and plan is not None:
        return [w for job in _table_jobs(op, plan, C) for w in _written(job, C)]
    if k == C.OP_CONV:
        return [(op.p[5], True)] + ([(op.p[6], False)] if op.p[7] else [])       # + backward-statistics rows
    if k == C.OP_CONV_SUM:
        return [(op.p[10], True), (op.p[7], True)]        # the residual sum written on the side + the conv output
    if k in (C.OP_SUM_TERMS, C.OP_BILINEAR_CAT):
        return [(op.p[0], True)]
    if k == C.OP_IM2COL_STEM or k == C.OP_NCHW_TO_NHWC:
        return [(op.p[1], True)]
    if k == C.OP_GRAD_TERM:
        return [(op.p[0], True)] + ([(op.p[7], True)] if op.p[7] else [])
    if k == C.OP_BILINEAR_CAT_BWD:
        return [(op.p[1 + j], True) for j in range(op.i[1])]
    if k == C.OP_HEAD_MIX:
        return [(op.p[3], True)]          # the head's raw output (its statistics feed a finalize launch)
    if k == C.OP_UPSAMPLE_T:
        return [(op.p[1 + j], True) for j in range(op.i[5])]
    if k == C.OP_BN_BWD_REDUCE:
        return [(op.p[6], True)] if op.p[6] else []           # the pooled, masked gradient kept for the apply pass
    if k == C.OP_POOL_REDUCE:
        return [(op.p[3 + 3 * l], True) for l in range(op.i[5])]
    if k == C.OP_HEAD_BWD:
        return [(op.p[3], op.i[6] == 2)]      # mode 1: statistics rows (a kept f32 tensor), mode 2: the gradient of raw y
    if k == C.OP_BN_FINALIZE:
        return [(op.p[6], False), (op.p[7], False), (op.p[8], False), (op.p[9], False)]
    if k == C.OP_BN_BWD_FINALIZE:
        return [(op.p[6], False)]
    if k == C.OP_BWD_FUSED:
        return [(op.p[7], True)] + ([(op.p[9], False)] if op.p[9] else [])       # dx + backward-statistics rows
    return []


@pytest.mark.parametrize('fused', ['fused_narrow', 'unfused'])
def test_bf16_op_by_op_with_exact_inputs_against_fp32_device_path(fused, monkeypatch):
    """The bf16 fidelity test that can fail. End to end, a randomly initialised 70-conv BatchNorm stack is
    chaotic (a perturbation grows ~1.2x per BatchNorm layer - measured: the fp32 path's gradient has cosine
    0.9998 against fp64, i.e. 1e-7 rounding becomes 2e-2), so bf16's 2^-9 roundings decorrelate the final
    gradient on ANY implementation (PyTorch's CPU bf16 kernels: cosine 0.28; this path: 0.34) and a whole-network
    cosine cannot tell a correct bf16 kernel from a wrong one. Here every op of the recorded forward and backward
    programs runs in bf16 on EXACT inputs - the fp32 device path's own activations / gradients / BatchNorm
    coefficients, rounded once - and its output is compared with the fp32 path's before being replaced by it.
    One wrong tile, tap, mask or coefficient in any of the ~1500 ops shows up at that op."""
    from hipnet import _capi as C
    from hipnet import synth
    # the two dtypes must record the same op lists: the fp32 instantiation of the fused block backward serves
    # 32-channel layers only, so the fused run restricts both to those (the 64-channel bf16 instantiation is
    # covered by tests/test_bwd_fused_gpu.py and by test_fused_backward_matches_unfused_backward below)
    # ... and the LDS-ring convolutions are bf16 only: the wide layers they serve keep their residual sums as separate
    # launches, so one run records both dtypes without fused sums (ring kernels op by op), the other without the ring
    # (hrnet_conv2d_sum op by op)
    monkeypatch.setenv('HRNET_MEASURE', '1')       # (measurement switches are ignored without it)
    if fused == 'unfused':
        monkeypatch.setenv('HRNET_FUSED_BWD', '0')
        ring_prev = C.call('hrnet_conv_ring_enable', 0)      # (the library reads HRNET_CONV_RING once: use the switch)
    else:
        ring_prev = C.call('hrnet_conv_ring_enable', 1)
        monkeypatch.setenv('HRNET_FUSE_SUM', '0')
        monkeypatch.setenv('HRNET_FUSED_MAXC', '32')
        monkeypatch.setenv('HRNET_FUSED_PW', '0')     # (the pointwise fused backward of layer1 is bf16 only: tests/test_bwd_pw_gpu.py)
    m32, sd = _model('fp32', init='reference', salt=4)
    m16, _ = _model('bf16', sd)
    batch = synth.rhd_batch(4, seed=21, img_h=128, img_w=128)
    x = torch.from_numpy(batch['imgs']).cuda()
    gt = torch.from_numpy(batch['heatmaps']).cuda()
    plans, outs = {}, {}
    for name, m, dt in (('32', m32, torch.float32), ('16', m16, torch.bfloat16)):
        m.train()
        net = m.hip()
        with torch.no_grad():
            net.pack_weights(for_backward=True)
        p = net.plan(4, 128, 128, True, True)
        hm = torch.empty((4, p.nj, p.out_act.H, p.out_act.W), device='cuda')
        inter = torch.empty((4, p.inter_act.C, p.inter_act.H, p.inter_act.W), device='cuda')
        p.fwd.set_ptr(p.in_op, 0, x.data_ptr())
        p.fwd.set_ptr(p.out_op, 1, hm.data_ptr())
        p.fwd.set_ptr(p.inter_op, 1, inter.data_ptr())
        plans[name], outs[name] = p, (hm, inter, dt, net)
    p32, p16 = plans['32'], plans['16']
    assert (p32.n_fused_blocks > 0) == (fused != 'unfused') and p32.n_fused_blocks == p16.n_fused_blocks
    assert p32.n_head_mix == 1 and p16.n_head_mix == 1      # the head without its concat, in both dtypes
    a32, k32 = _ptr_maps(p32, torch.float32)
    a16, k16 = _ptr_maps(p16, torch.bfloat16)

    stat_errs = []

    def lockstep(prog32, prog16, tag):
        assert len(prog32) == len(prog16)
        errs = []
        for k in range(len(prog32)):
            o32, o16 = prog32._arr[k], prog16._arr[k]
            assert int(o32.kind) == int(o16.kind), (tag, k)
            if int(o32.kind) in (C.OP_EVENT_RECORD, C.OP_STREAM_WAIT):
                continue
            prog32.run(k, k + 1)
            prog16.run(k, k + 1)
            if int(o32.kind) == C.OP_BN_FINALIZE_TABLE:
                # the arrays the backward pass reads (scale, shift, mean, invstd of every BatchNorm): exact values
                for name, b32 in p32.bns.items():
                    b16 = p16.bns[name]
                    for t32, t16 in ((b32.scale, b16.scale), (b32.shift, b16.shift), (b32.mean, b16.mean), (b32.invstd, b16.invstd)):
                        t16.copy_(t32)
                continue
            for (q32, is_act), (q16, _) in zip(_written(o32, C, p32), _written(o16, C, p16)):
                if is_act:
                    if q32 not in a32:
                        continue                      # the NCHW outputs (compared below)
                    kind, i = a32[q32]
                    assert a16[q16] == (kind, i)
                    v32 = _act_view(p32, kind, i, torch.float32)
                    v16 = _act_view(p16, kind, i, torch.bfloat16)
                    name = ('d ' if kind == 'g' else '') + p32.acts[i].name
                else:
                    v32, v16 = k32[q32], k16[q16]
                    name = 'coef@{}'.format(k)
                    if v32.numel() != v16.numel():
                        continue
                    if int(o32.kind) in (C.OP_CONV, C.OP_BWD_FUSED):
                        # backward-statistics rows (sum dz, sum dz*y per channel): compared as vectors over the
                        # channels, then replaced like every other output (the consumer - a finalize op, or the next
                        # fused launch, which builds its coefficients from the rows itself - sees exact sums).
                        # Many of these sums cancel structurally (the gradient behind a BatchNorm sums to zero per
                        # channel; what is left are border and ReLU-mask terms), so the 2^-9 rounding of the inputs
                        # is large against them: they are held to a loose band.
                        c = p32.acts[a32[o32.p[5] if int(o32.kind) == C.OP_CONV else o32.p[7]][1]].C
                        s32, s16 = v32.view(-1, 2, c).double().sum(0), v16.view(-1, 2, c).double().sum(0)
                        for w in range(2):
                            stat_errs.append(((s32[w] - s16[w]).norm().item() / max(s32[w].norm().item(), 1e-30),
                                              int(o32.kind), k, 'sum dz' + ('*y' if w else '') + ' behind op {}'.format(k)))
                        v16.copy_(v32)
                        continue
                d = (v16.float() - v32).double().norm().item()
                n = v32.double().norm().item()
                errs.append((d / max(n, 1e-30), int(o32.kind), k, name))
                v16.copy_(v32)                        # teacher forcing: the next op sees the exact value, rounded once
        return errs

    fe = lockstep(p32.fwd, p16.fwd, 'fwd')
    torch.cuda.synchronize()
    hm32, hm16 = outs['32'][0], outs['16'][0]
    # backward: one upstream gradient for both (d HeatmapLoss / d heat maps of the fp32 path)
    g_hm = ((hm32 - gt) * (2.0 / (hm32.shape[0] * hm32.shape[1]))).contiguous()
    for name in ('32', '16'):
        net = outs[name][3]
        net.prepare_grads()
        plans[name].bwd.set_ptr(plans[name].gout_op, 0, g_hm.data_ptr())
    be = lockstep(p32.bwd, p16.bwd, 'bwd')
    torch.cuda.synchronize()

    def report(errs, what):
        errs = sorted(errs, reverse=True)
        print('{}: {} compared outputs, median {:.2e}, p99 {:.2e}, worst: {}'.format(
            what, len(errs), float(np.median([e[0] for e in errs])), float(np.percentile([e[0] for e in errs], 99)),
            ['{:.2e} {} (op {} kind {})'.format(e[0], e[3], e[2], e[1]) for e in errs[:4]]))
        return errs
    fe, be = report(fe, 'forward'), report(be, 'backward')
    se = report(stat_errs, 'backward-statistics sums')
    ge_w, ge_b = [], []
    g32 = {k: p.grad.detach().double() for k, p in m32.named_parameters()}
    g16 = {k: p.grad.detach().double() for k, p in m16.named_parameters()}
    gmax = max(v.abs().max().item() for v in g32.values())
    for k, ref in g32.items():
        if ref.abs().max().item() < 1e-6 * gmax:
            continue
        (ge_w if ref.dim() == 4 else ge_b).append(((g16[k] - ref).norm().item() / ref.norm().item(), 0, 0, k))
    ge_w = report(ge_w, 'conv weight gradients')
    ge_b = report(ge_b, 'BatchNorm / bias gradients (per-channel sums)')
    assert (hm16 - hm32).norm().item() <= 1e-2 * hm32.norm().item()
    assert len(fe) >= 400 and len(be) >= (900 if fused == 'unfused' else 600) and len(ge_w) >= 300
    # bf16 operands (2^-9 relative rounding of inputs, weights and the stored result), f32 accumulation.
    # Measured (MI355X): forward median 1.6e-3 / worst 4.3e-3; backward median 2.4e-3; conv weight gradients
    # p99 4.4e-3; per-channel sums up to 5e-2 where the sum cancels (see above).
    assert fe[0][0] <= 1e-2, fe[0]
    assert float(np.median([e[0] for e in fe])) <= 4e-3
    assert be[0][0] <= 8e-2 and float(np.percentile([e[0] for e in be], 99)) <= 5e-2, be[0]   # worst: BN-backward outputs that cancel
    assert float(np.median([e[0] for e in be])) <= 6e-3
    assert ge_w[0][0] <= 1.5e-2, ge_w[0]
    assert float(np.median([e[0] for e in ge_w])) <= 5e-3
    assert ge_b[0][0] <= 0.15 and float(np.median([e[0] for e in ge_b])) <= 1e-2, ge_b[0]
    assert se[0][0] <= 0.3 and float(np.median([e[0] for e in se])) <= 2e-2, se[0]


@pytest.mark.parametrize('dtype', ['fp32', 'bf16'])
def test_fused_backward_matches_unfused_backward(dtype, monkeypatch):
    """the fused block backward (hrnet_conv3x3_bwd_fused) against the unfused sequence (grad_term + wgrad +
    conv_bs) in the SAME dtype: the forward programs are identical, so activations and ReLU masks agree bit for
    bit and the backward - linear in the upstream gradient - differs by rounding order only: no chaos here, every
    parameter gradient is compared tightly. bf16 covers both fused instantiations (32 and 64 channels)."""
    from hipnet import synth
    batch = synth.rhd_batch(4, seed=31, img_h=128, img_w=128)
    monkeypatch.setenv('HRNET_DETERMINISTIC', '1')     # bit-reproducible batch statistics: the two forward passes agree exactly
    monkeypatch.setenv('HRNET_MEASURE', '1')           # (measurement switches are ignored without it)
    monkeypatch.setenv('HRNET_FUSED_BWD', '0')
    mu, sd = _model(dtype, init='reference', salt=6)
    hm_u, _, loss_u, gu = _hip_step(mu, batch)
    assert mu.hip().plan(4, 128, 128, True, True).n_fused_blocks == 0
    monkeypatch.setenv('HRNET_FUSED_BWD', '1')
    mf, _ = _model(dtype, sd)
    hm_f, _, loss_f, gf = _hip_step(mf, batch)
    nf = mf.hip().plan(4, 128, 128, True, True).n_fused_blocks
    # w32: 32 BasicBlocks of 32 channels + 32 of 64 (the 64-channel block whose input also feeds transition2 stays
    # unfused) + in bf16 the four Bottlenecks of layer1 (hrnet_conv1x1_bwd_fused)
    assert nf == (32 if dtype == 'fp32' else 67), nf
    assert torch.equal(hm_u, hm_f) and loss_u == loss_f
    errs = sorted(((gf[k] - gu[k]).norm().item() / max(gu[k].norm().item(), 1e-30), k) for k in gu
                  if gu[k].abs().max().item() > 0)
    cos = _cos(gf, gu)
    print('{} fused vs unfused backward: cosine {:.7f}, per-tensor rel L2 median {:.2e} max {:.2e} ({})'.format(
        dtype, cos, float(np.median([e[0] for e in errs])), errs[-1][0], errs[-1][1]))
    if dtype == 'fp32':
        assert cos >= 0.999999 and errs[-1][0] <= 2e-3 and float(np.median([e[0] for e in errs])) <= 1e-4
    else:
        assert cos >= 0.999 and errs[-1][0] <= 0.15 and float(np.median([e[0] for e in errs])) <= 1.5e-2
