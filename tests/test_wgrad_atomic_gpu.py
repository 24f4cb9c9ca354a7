"""The float-atomic weight-gradient forms (HR_OP_WGRAD i[12] = 1, HR_OP_BWD_FUSED / HR_OP_BWD_PW i[8] = 1: every
workgroup ADDS its tile straight into the OIHW f32 gradient) against the slab forms (one f32 slab per split +
hrnet_wgrad_reduce): the same products summed in another order, so the two agree to f32 rounding - and the atomic
form must ADD to what the gradient buffer already holds (PyTorch's .grad accumulation).
Also: the batched element-wise launches (HR_OP_EW_TABLE) against one launch per job, bit for bit, inside the
recorded backward program. Reference ops: autograd of nn.Conv2d / nn.BatchNorm2d in lib/models/pose_hrnet.py:22-57."""
import ctypes

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DT = torch.bfloat16


def _h():
    import hip_helpers as hh
    return hh


def _C():
    from hipnet import _capi as C
    return C


def _run(op):
    C = _C()
    C.call('hrnet_program_run', ctypes.byref(op), 1, C.stream_ptr())


WG_CASES = [
    # N, H, W, Cin (tensor), Cout (tensor), ks, stride, real Cin, real Cout
    (6, 32, 32, 32, 32, 3, 1, 32, 32),
    (5, 16, 16, 128, 128, 3, 1, 128, 128),     # 16x16 tiles, four input-channel blocks, LDS transposition in four passes
    (9, 8, 8, 256, 256, 3, 1, 256, 256),       # 8x8 tiles: the tile transposition needs more LDS than the operand images
    (4, 32, 32, 32, 64, 3, 2, 32, 64),         # stride 2 (fuse-layer down path)
    (3, 64, 64, 480, 32, 1, 1, 480, 21),       # the head's last layer: 21 real output channels in a 32-channel tensor
    (4, 32, 32, 64, 32, 1, 1, 64, 32),         # fuse-layer 1x1
    (2, 24, 20, 48, 96, 3, 1, 48, 96),         # w48 widths: ragged channel blocks
]


@pytest.mark.parametrize('case', WG_CASES)
def test_wgrad_atomic_equals_slabs_plus_reduce(case):
    hh, C = _h(), _C()
    N, H, W, Cin, Cout, ks, stride, cin_r, cout_r = case
    pad = ks // 2
    Ho, Wo = (H + 2 * pad - ks) // stride + 1, (W + 2 * pad - ks) // stride + 1
    g = torch.Generator().manual_seed(3 + Cin + Cout + ks)
    x = torch.randn(N, H, W, Cin, generator=g).to(DT).to(hh.DEV)
    dy = torch.randn(N, Ho, Wo, Cout, generator=g).to(DT).to(hh.DEV)
    if cout_r < Cout:
        dy[..., cout_r:] = 0
    sc = (torch.rand(Cin, generator=g) + 0.5).to(hh.DEV)
    sh = (torch.rand(Cin, generator=g) - 0.5).to(hh.DEV)
    ref = hh.wgrad(x, dy, N, H, W, Cin, Ho, Wo, Cout, ks, stride, DT, in_scale=sc, in_shift=sh, in_relu=True,
                   cout_real=cout_r, cin_real=cin_r)
    ns = C.call('hrnet_wgrad_splits', 1, N, Ho, Wo, Cout, Cin, ks, stride)
    base = torch.randn(cout_r, cin_r, ks, ks, generator=g).to(hh.DEV)         # what .grad already holds
    got = base.clone()
    op = C.HrOp()
    op.kind = C.OP_WGRAD
    for k, v in enumerate((1, N, H, W, Cin, Ho, Wo, Cout, ks, stride, 1, ns, 1, cout_r, cin_r)):
        op.i[k] = v
    for k, t in enumerate((x, dy, sc, sh, got)):
        op.p[k] = t.data_ptr()
    _run(op)
    hh.sync()
    d = (got - base - ref).abs().max().item()
    assert d <= 2e-5 * max(1.0, ref.abs().max().item()), (d, ref.abs().max().item(), ns)


FUSED_CASES = [
    # N, H, W, Cin, Cout
    (6, 32, 32, 32, 32),
    (40, 64, 64, 32, 32),      # several tiles per workgroup
    (5, 32, 32, 64, 64),       # two input-channel blocks per walk
    (3, 20, 24, 48, 48),       # ragged blocks (w48)
    (3, 16, 16, 64, 32),
]


@pytest.mark.parametrize('case', FUSED_CASES)
def test_fused_backward_atomic_weight_gradient_equals_slabs(case):
    """hrnet_conv3x3_bwd_fused with the weight-gradient tiles added into the OIHW gradient (through LDS, 16 output
    channels at a time) against its slab form + hrnet_wgrad_reduce; dx and the statistics rows must not change"""
    hh, C = _h(), _C()
    N, H, W, Cin, Cout = case
    g = torch.Generator().manual_seed(9 + Cin + Cout + N)
    x = torch.randn(N, H, W, Cin, generator=g).to(DT).to(hh.DEV)
    dz = torch.randn(N, H, W, Cout, generator=g).to(DT).to(hh.DEV)
    y = torch.randn(N, H, W, Cout, generator=g).to(DT).to(hh.DEV)
    bsy = torch.randn(N, H, W, Cin, generator=g).to(DT).to(hh.DEV)
    coef = torch.randn(3 * Cout, generator=g).to(hh.DEV) * 0.5
    sc = (torch.rand(Cin, generator=g) + 0.5).to(hh.DEV)
    sh = (torch.rand(Cin, generator=g) - 0.5).to(hh.DEV)
    w = torch.randn(Cout, Cin, 3, 3, generator=g) / np.sqrt(Cin * 9)
    wT, _, _ = hh.pack_weights(w, DT, mode=1)
    ns = C.call('hrnet_bwd_fused_splits', 1, N, H, W, Cin, Cout)
    outs = []
    for atomic in (0, 1):
        dx = torch.full((N, H, W, Cin), float('nan'), dtype=DT, device=hh.DEV)
        rows = torch.zeros(ns, 2, Cin, device=hh.DEV)
        if atomic:
            acc = torch.zeros(Cout, Cin, 3, 3, device=hh.DEV)
            dst = acc
        else:
            slabs = torch.zeros(ns, Cout, 9, Cin, device=hh.DEV)
            dst = slabs
        op = C.HrOp()
        op.kind = C.OP_BWD_FUSED
        for k, v in enumerate((1, N, H, W, Cin, Cout, 1, 1, atomic, Cout, Cin)):
            op.i[k] = v
        for k, t in enumerate((dz, y, coef, x, sc, sh, wT, dx, None, rows, bsy, dst)):
            op.p[k] = C.ptr(t)
        _run(op)
        if not atomic:
            acc = torch.zeros(Cout, Cin, 3, 3, device=hh.DEV)
            C.call('hrnet_wgrad_reduce', slabs.data_ptr(), acc.data_ptr(), ns, Cout, Cin, 3, Cout, Cin, 0, 0, C.stream_ptr())
        hh.sync()
        outs.append((dx, rows.clone(), acc))
    (dx0, r0, w0), (dx1, r1, w1) = outs
    assert torch.equal(dx0.view(torch.int16), dx1.view(torch.int16)) and torch.equal(r0, r1)
    assert float((w0 - w1).abs().max()) <= 2e-5 * max(1.0, float(w0.abs().max()))


def _step(monkeypatch, env):
    from hipnet import synth
    from core.loss import HeatmapLoss
    import test_bench_path_gpu as T
    monkeypatch.setenv('HRNET_MEASURE', '1')       # (HRNET_BATCH_SUM* are measurement switches: ignored without it)
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    m, _ = T._model('bf16', init='reference', salt=3)
    m.train()
    b = synth.rhd_batch(4, seed=5, img_h=128, img_w=128)
    x, gt = torch.from_numpy(b['imgs']).cuda(), torch.from_numpy(b['heatmaps']).cuda()
    hm, _ = m(x)
    HeatmapLoss()(hm, gt).backward()
    torch.cuda.synchronize()
    net = m.hip()
    return hm.detach().cpu(), net.flat_g.detach().cpu().clone(), net.plan(4, 128, 128, True, True), m


def test_training_step_atomic_weight_gradients_match_ordered_slab_sums(monkeypatch):
    """whole recorded backward, ordered batch statistics in both runs (identical activations), weight gradients once
    by atomics into the OIHW gradients and once through slabs + ordered sums: every parameter gradient agrees to
    f32 summation noise"""
    hm_a, g_a, plan_a, m_a = _step(monkeypatch, {'HRNET_DETERMINISTIC': '1', 'HRNET_WGRAD_ATOMIC': '1'})
    hm_s, g_s, plan_s, m_s = _step(monkeypatch, {'HRNET_DETERMINISTIC': '1', 'HRNET_WGRAD_ATOMIC': '0'})
    assert plan_a.wgrad_atomic and not plan_s.wgrad_atomic
    assert plan_a.slab_bytes < 0.05 * plan_s.slab_bytes          # only the stem keeps a slab
    assert torch.equal(hm_a, hm_s)
    net = m_a.hip()
    worst = 0.0
    for p in net.params:
        off, n = net.offsets[id(p)]
        a, s = g_a[off:off + n], g_s[off:off + n]
        if float(s.abs().max()) > 0:
            worst = max(worst, float((a - s).norm() / s.norm()))
    assert worst <= 2e-5, worst


def test_batched_sum_backward_launches_change_nothing(monkeypatch):
    """HR_OP_EW_TABLE (the fuse sums' reduce / finalize / apply passes of a HighResolutionModule as three launches)
    runs the same device code on the same operands as one launch per pass: bit-identical gradients"""
    C = _C()
    hm_b, g_b, plan_b, _ = _step(monkeypatch, {'HRNET_DETERMINISTIC': '1', 'HRNET_BATCH_SUMBWD': '1'})
    hm_u, g_u, plan_u, _ = _step(monkeypatch, {'HRNET_DETERMINISTIC': '1', 'HRNET_BATCH_SUMBWD': '0'})
    nb = sum(1 for o in plan_b.bwd.ops if int(o.kind) == C.OP_EW_TABLE)
    nu = sum(1 for o in plan_u.bwd.ops if int(o.kind) == C.OP_EW_TABLE)
    assert nb >= 16 and nu == 0 and plan_b.n_batched_jobs >= 150, (nb, nu, plan_b.n_batched_jobs)
    assert len(plan_b.bwd) < len(plan_u.bwd) - 150
    assert torch.equal(hm_b, hm_u) and torch.equal(g_b, g_u)


def test_batched_forward_sums_change_nothing(monkeypatch):
    """the sums of a HighResolutionModule's outputs as ONE table-driven launch (HR_OP_EW_TABLE of HR_OP_SUM_TERMS jobs)
    run the same device code on the same operands as one launch per sum: bit-identical heat maps and gradients -
    with per-BatchNorm finalize launches (deterministic) and with BatchNorm-from-sums in the prologue (default)"""
    C = _C()
    for det in ('1', '0'):
        env = {'HRNET_DETERMINISTIC': '1'} if det == '1' else {'HRNET_DETERMINISTIC': '0'}
        hm_b, g_b, plan_b, _ = _step(monkeypatch, dict(env, HRNET_BATCH_SUMFWD='1'))
        hm_u, g_u, plan_u, _ = _step(monkeypatch, dict(env, HRNET_BATCH_SUMFWD='0'))
        nb = sum(1 for o in plan_b.fwd.ops if int(o.kind) == C.OP_EW_TABLE)
        nu = sum(1 for o in plan_u.fwd.ops if int(o.kind) == C.OP_EW_TABLE)
        assert nb == 8 and nu == 0 and plan_b.n_batched_fwd_sums == 26, (nb, nu, plan_b.n_batched_fwd_sums)
        if det == '1':
            assert torch.equal(hm_b, hm_u) and torch.equal(g_b, g_u)
        else:
            # batch statistics by float atomics: two runs differ in summation order, and a randomly initialised
            # 70-BatchNorm stack amplifies that (test_bench_path_gpu.py: whole-network band 0.3-0.45 for bf16)
            assert torch.isfinite(hm_b).all() and float((hm_b - hm_u).norm() / hm_u.norm()) <= 0.45


def test_atomic_batch_statistics_differ_from_ordered_ones_by_f32_rounding_only(monkeypatch):
    """the default training forward adds a conv's batch sums into 8 partial copies with float atomics
    (hr_bn_from_sums: f64 combination, mean / E[x^2] - mean^2); HRNET_DETERMINISTIC=1 sums per-workgroup rows in a
    fixed order. On identical inputs (the stem's first BatchNorm: its conv sees the same image and weights in both runs)
    the two statistics agree to f32 summation noise - the gap between the two modes further down the network is the
    amplification of that noise, not a different estimator"""
    _, _, plan_a, m_a = _step(monkeypatch, {'HRNET_DETERMINISTIC': '0'})
    mean_a, inv_a = plan_a.bns['bn1'].mean.clone(), plan_a.bns['bn1'].invstd.clone()
    assert plan_a.bn_sums
    _, _, plan_d, m_d = _step(monkeypatch, {'HRNET_DETERMINISTIC': '1'})
    assert not plan_d.bn_sums
    mean_d, inv_d = plan_d.bns['bn1'].mean, plan_d.bns['bn1'].invstd
    sd = 1.0 / inv_d
    assert float(((mean_a - mean_d).abs() / sd).max()) <= 2e-6          # in units of the channel's standard deviation
    assert float(((inv_a - inv_d).abs() / inv_d).max()) <= 2e-6
