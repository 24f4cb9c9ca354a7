"""GPU parity of the LDS-ring 3x3 convolutions (csrc/conv_ring.hip) that hrnet_conv2d / hrnet_conv2d_bnref /
hrnet_conv2d_bwdstats route the branch layers to (BasicBlock.conv1 / conv2, reference
lib/models/pose_hrnet.py:41-57, and their input gradients).

Every case asserts through hrnet_conv_ring_supported() that the ring instantiation IS what runs, then checks
  * against plain torch fp32 on the CPU (F.conv2d of the same bf16-rounded operands, tolerance 3e-2 relative to the
    largest output: the bf16 bar of test_kernels_gpu.py), image by image;
  * against the tile-walking body of conv_body.h with the routing switched off (hrnet_conv_ring_enable(0)): the two
    bodies accumulate in the same order, so the outputs must agree BIT FOR BIT; the statistics (float atomics /
    rows of a different grid) to f32 summation noise.
Shapes: the four branch widths, tiles that overhang the map, image counts that leave the last multi-image tile
partly empty, several tiles per workgroup (the benchmark's walk), input read raw / through BatchNorm batch sums /
through scale-shift arrays."""
import ctypes

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
DT = torch.bfloat16
TOL = 3e-2


def _h():
    import hip_helpers as hh
    return hh


def _C():
    from hipnet import _capi as C
    return C


def _q(t):
    return t.to(DT).float()


@pytest.fixture(autouse=True)
def _ring_on():
    C = _C()
    prev = C.call('hrnet_conv_ring_enable', 1)
    yield
    C.call('hrnet_conv_ring_enable', 1 if prev != 0 else 0)


FWD_CASES = [
    # N, H, W, Cin, Cout, input mode
    (3, 20, 37, 32, 32, 'sums'),       # tiles overhang the map on both axes
    (2, 16, 16, 32, 32, 'raw'),
    (70, 64, 64, 32, 32, 'sums'),      # > 512 tile-workgroups: two or three tiles per workgroup, ring wraps
    (2, 24, 50, 64, 64, 'sums'),       # two K chunks with resident weights
    (40, 32, 32, 64, 64, 'arrays'),
    (64, 32, 32, 64, 64, 'raw'),
    (3, 20, 37, 128, 64, 'sums'),      # streamed weights, one output-channel block, overhanging tiles
    (9, 16, 16, 128, 128, 'raw'),      # two output-channel blocks
    (64, 16, 16, 128, 128, 'sums'),    # the benchmark's third branch
    (5, 8, 8, 256, 256, 'sums'),       # multi-image tiles: the last tile holds one image of four
    (6, 8, 8, 128, 192, 'arrays'),     # w48-like channel counts (three output-channel blocks)
    (64, 8, 8, 256, 256, 'raw'),       # the benchmark's fourth branch
    (3, 8, 8, 384, 384, 'sums'),       # w48's widest branch: the 27 KB staging image of its BatchNorm sums fits the 40 KB slot
]


@pytest.mark.parametrize('case', FWD_CASES)
def test_ring_forward_matches_cpu_and_the_tile_walking_body(case):
    hh, C = _h(), _C()
    N, H, W, Cin, Cout, mode = case
    C.call('hrnet_conv_ring_enable', 2)      # (2: the wide 16x16-tile instantiation on maps larger than 16x16 as well)
    assert C.call('hrnet_conv_ring_supported', 1, N, H, W, Cin, Cout) > 0
    g = torch.Generator().manual_seed(7 + N + Cin + Cout)
    x = _q(torch.randn(N, Cin, H, W, generator=g) * 1.5 + 0.3)
    w = _q(torch.randn(Cout, Cin, 3, 3, generator=g) / np.sqrt(Cin * 9))
    gamma = torch.rand(Cin, generator=g) + 0.5
    beta = torch.rand(Cin, generator=g) - 0.5
    eps = 1e-5
    # batch statistics of x exactly as the producer's epilogue would have summed them, spread over the 8 copies
    s1 = x.double().sum((0, 2, 3))
    s2 = (x.double() ** 2).sum((0, 2, 3))
    frac = torch.rand(8, 1, generator=g).double()
    frac /= frac.sum()
    sums = torch.stack([frac * s1[None], frac * s2[None]], 1).float().contiguous()
    cnt = float(N * H * W)
    s1f, s2f = sums[:, 0].double().sum(0), sums[:, 1].double().sum(0)
    mean = s1f / cnt
    var = (s2f / cnt - mean * mean).clamp_min(0)
    invstd = 1.0 / torch.sqrt(var.float() + eps)
    scale = gamma * invstd
    shift = beta - mean.float() * scale
    if mode == 'raw':
        xa = x
    else:
        xa = _q(F.relu(x * scale.view(1, -1, 1, 1) + shift.view(1, -1, 1, 1)))
    ref = F.conv2d(xa, w, None, stride=1, padding=1)
    wp, cop, cip = hh.pack_weights(w, DT)
    assert cop == Cout and cip == Cin
    xd = hh.nhwc(x, DT)
    gb = torch.cat([gamma, beta]).to(hh.DEV).contiguous()
    sums_d, scale_d, shift_d = sums.to(hh.DEV), scale.to(hh.DEV), shift.to(hh.DEV)
    outs = []
    for ring in (2, 0):
        C.call('hrnet_conv_ring_enable', ring)
        y = torch.full((N, H, W, Cout), float('nan'), dtype=DT, device=hh.DEV)
        st = torch.zeros(8, 2, Cout, dtype=torch.float32, device=hh.DEV)
        if mode == 'sums':
            C.call('hrnet_conv2d_bnref', 1, xd.data_ptr(), wp.data_ptr(), sums_d.data_ptr(), gb.data_ptr(),
                   gb.data_ptr() + 4 * Cin, 1.0 / cnt, eps, None, y.data_ptr(), st.data_ptr(), N, H, W, Cin, H, W, Cout, 3, 1, 1,
                   C.stream_ptr())
        elif mode == 'raw':
            C.call('hrnet_conv2d_bnref', 1, xd.data_ptr(), wp.data_ptr(), None, None, None, 0.0, 0.0, None, y.data_ptr(),
                   st.data_ptr(), N, H, W, Cin, H, W, Cout, 3, 1, 0, C.stream_ptr())
        else:
            # eval-style launch: precomputed scale / shift arrays, no output statistics
            C.call('hrnet_conv2d', 1, xd.data_ptr(), wp.data_ptr(), scale_d.data_ptr(), shift_d.data_ptr(), None, y.data_ptr(), None,
                   N, H, W, Cin, H, W, Cout, 3, 1, 0, 1, 0, C.stream_ptr())
        hh.sync()
        outs.append((y, st.double().sum(0).cpu()))
    C.call('hrnet_conv_ring_enable', 1)
    (y1, st1), (y0, st0) = outs
    got = hh.from_nhwc(y1, Cout)
    assert not torch.isnan(got).any()
    assert hh.rel_err(got, ref) <= TOL
    per_img = (got - ref).abs().amax((1, 2, 3)) / ref.abs().amax()
    assert float(per_img.max()) <= TOL
    if H == 8:
        # (the tile-walking body takes 64 input channels per stage on 8x8 maps, the ring 32: another summation order)
        assert hh.rel_err(y1.float().cpu(), y0.float().cpu()) <= 4e-3
    else:
        assert torch.equal(y1.view(torch.int16), y0.view(torch.int16)), 'ring and tile-walking bodies differ'
    if mode != 'arrays':
        ref_s1, ref_s2 = ref.double().sum((0, 2, 3)), (ref.double() ** 2).sum((0, 2, 3))
        scale1 = ref.double().abs().sum((0, 2, 3)).max().item()
        assert float((st1[0] - ref_s1).abs().max()) <= (2 * TOL + 1e-5) * scale1
        assert hh.rel_err(st1[1], ref_s2) <= 5 * TOL
        assert float((st1 - st0).abs().max() / st0.abs().max()) <= (1e-5 if H != 8 else 2e-3)


def test_ring_declines_shapes_whose_batchnorm_staging_does_not_fit_a_slot():
    """The prologue stages [8][2][Cin] batch sums + gamma | beta in the last ring slot: 27 KB for Cin = 384, more than the
    20.7 KB slot of the 16x16-tile instantiation - such a launch used to spill into weight slot 0. The planner now leaves
    it to the tile-walking body (and the launch through the dispatcher stays correct)."""
    hh, C = _h(), _C()
    C.call('hrnet_conv_ring_enable', 2)
    assert C.call('hrnet_conv_ring_supported', 1, 2, 16, 16, 384, 384) == 0
    assert C.call('hrnet_conv_ring_supported', 1, 2, 16, 16, 256, 256) > 0
    assert C.call('hrnet_conv_ring_supported', 1, 2, 8, 8, 384, 384) > 0
    N, H, W, Cc = 2, 16, 16, 384
    g = torch.Generator().manual_seed(5)
    x = _q(torch.randn(N, Cc, H, W, generator=g))
    w = _q(torch.randn(Cc, Cc, 3, 3, generator=g) / np.sqrt(Cc * 9))
    gamma, beta = torch.rand(Cc, generator=g) + 0.5, torch.rand(Cc, generator=g) - 0.5
    s1, s2 = x.double().sum((0, 2, 3)), (x.double() ** 2).sum((0, 2, 3))
    sums = torch.zeros(8, 2, Cc)
    sums[3, 0], sums[3, 1] = s1.float(), s2.float()
    cnt = float(N * H * W)
    mean = s1 / cnt
    invstd = 1.0 / torch.sqrt((s2 / cnt - mean * mean).clamp_min(0).float() + 1e-5)
    scale = gamma * invstd
    shift = beta - mean.float() * scale
    ref = F.conv2d(_q(F.relu(x * scale.view(1, -1, 1, 1) + shift.view(1, -1, 1, 1))), w, None, stride=1, padding=1)
    wp, _, _ = hh.pack_weights(w, DT)
    xd, gb, sd = hh.nhwc(x, DT), torch.cat([gamma, beta]).to(hh.DEV).contiguous(), sums.to(hh.DEV)
    y = torch.full((N, H, W, Cc), float('nan'), dtype=DT, device=hh.DEV)
    st = torch.zeros(8, 2, Cc, dtype=torch.float32, device=hh.DEV)
    C.call('hrnet_conv2d_bnref', 1, xd.data_ptr(), wp.data_ptr(), sd.data_ptr(), gb.data_ptr(), gb.data_ptr() + 4 * Cc,
           1.0 / cnt, 1e-5, None, y.data_ptr(), st.data_ptr(), N, H, W, Cc, H, W, Cc, 3, 1, 1, C.stream_ptr())
    hh.sync()
    C.call('hrnet_conv_ring_enable', 1)
    assert hh.rel_err(hh.from_nhwc(y, Cc), ref) <= TOL


BS_CASES = [
    # N, H, W, C, mask source ('plain': no backward statistics at all), accumulate, store masked
    (3, 16, 16, 128, 'plain', 1, 0),     # an accumulating input gradient without statistics
    (5, 8, 8, 256, 'plain', 1, 0),
    (4, 16, 16, 128, 'plain', 0, 0),     # plain input gradient: the forward instantiation serves it
    (3, 16, 16, 128, 'none', 0, 0),
    (3, 16, 16, 128, 'affine', 1, 0),
    (5, 20, 24, 128, 'mask', 1, 1),      # overhanging 16x16 tiles
    (5, 8, 8, 256, 'mask', 0, 1),        # partly empty multi-image tile
    (7, 8, 8, 256, 'affine', 1, 0),
    (64, 16, 16, 128, 'mask', 1, 1),     # benchmark shapes
    (64, 8, 8, 256, 'affine', 0, 0),
]


@pytest.mark.parametrize('case', BS_CASES)
def test_ring_input_gradient_with_backward_statistics(case):
    """hrnet_conv2d_bwdstats on the wide branches: v = conv(dY, W^T) (+ old), dz = v * [m > 0], rows = (sum dz, sum dz*y)"""
    hh, C = _h(), _C()
    N, H, W, Cc, masked, accumulate, store_masked = case
    from hipnet._capi import HrOp
    g = torch.Generator().manual_seed(11 + N + Cc)
    dy = _q(torch.randn(N, Cc, H, W, generator=g))
    w = _q(torch.randn(Cc, Cc, 3, 3, generator=g) / np.sqrt(Cc * 9))          # forward weights [co][ci]
    bs_y = _q(torch.randn(N, Cc, H, W, generator=g))
    bs_m = _q(torch.randn(N, Cc, H, W, generator=g)) if masked == 'mask' else None
    bsc = (torch.rand(Cc, generator=g) + 0.5) if masked == 'affine' else None
    bsh = (torch.rand(Cc, generator=g) - 0.5) if masked == 'affine' else None
    old = _q(torch.randn(N, Cc, H, W, generator=g))
    # input gradient of y = conv(x, w): dx = conv_transpose(dy, w)
    v = F.conv_transpose2d(dy, w, stride=1, padding=1)
    if accumulate:
        v = v + old
    if masked == 'mask':
        m = bs_m
    elif masked == 'affine':
        m = bs_y * bsc.view(1, -1, 1, 1) + bsh.view(1, -1, 1, 1)
    else:
        m = None
    dz = v if m is None else v * (m > 0).float()
    ref_rows = torch.stack([dz.double().sum((0, 2, 3)), (dz.double() * bs_y.double()).sum((0, 2, 3))])
    ref_store = dz if store_masked else v
    wp, cop, cip = hh.pack_weights(w, DT, mode=1)
    dyd, byd = hh.nhwc(dy, DT), hh.nhwc(bs_y, DT)
    bmd = hh.nhwc(bs_m, DT) if bs_m is not None else None
    outs = []
    for ring in (2, 0):
        C.call('hrnet_conv_ring_enable', ring)
        if ring:
            assert C.call('hrnet_conv_ring_supported', 1, N, H, W, Cc, Cc) >= 3
        nrows = C.call('hrnet_conv_rows_bwdstats', 1, N, H, W, Cc, Cc, 3, 1)
        y = hh.nhwc(old, DT).clone()
        rows = torch.full((nrows, 2, Cc), float('nan'), dtype=torch.float32, device=hh.DEV)
        if masked == 'plain':
            rows, byd = None, None
        op = HrOp()
        op.kind = C.OP_CONV
        for k, val in enumerate((1, N, H, W, Cc, H, W, Cc, 3, 1, 0, 0, accumulate, 0, store_masked)):
            op.i[k] = val
        bscd = bsc.to(hh.DEV) if bsc is not None else None
        bshd = bsh.to(hh.DEV) if bsh is not None else None
        for k, t in ((0, dyd), (1, wp), (5, y), (6, rows), (7, byd), (8, bmd), (9, bscd), (10, bshd)):
            op.p[k] = C.ptr(t)
        C.call('hrnet_program_run', ctypes.byref(op), 1, C.stream_ptr())
        hh.sync()
        outs.append((y, rows.double().sum(0).cpu() if rows is not None else None, nrows))
    C.call('hrnet_conv_ring_enable', 1)
    (y1, r1, n1), (y0, r0, n0) = outs
    got = hh.from_nhwc(y1, Cc)
    assert not torch.isnan(got).any() and (r1 is None or not torch.isnan(r1).any())
    assert hh.rel_err(got, ref_store) <= TOL
    # the two bodies sum the taps and input-channel chunks in the same order where the tile-walking body takes its wide
    # tile: the same bits. On 8x8 tiles (the 8x8 maps, and maps far from a multiple of 16 - 20x24 here - since round 4:
    # choose_tile in csrc/conv_body.h) its order differs: bf16 rounding of the same sums
    t5 = (ctypes.c_int * 5)()
    C.call('hrnet_conv_tile_walk', N, H, W, Cc, 3, 1, 0 if masked == 'plain' else 1, 0, t5)
    small = t5[1] == 8
    assert small == (H == 8 or (H, W) == (20, 24))
    if small:
        assert hh.rel_err(y1.float().cpu(), y0.float().cpu()) <= 4e-3
    else:
        assert torch.equal(y1.view(torch.int16), y0.view(torch.int16)), 'ring and tile-walking bodies differ'
    if masked == 'plain':
        return
    scale = dz.double().abs().sum((0, 2, 3)).max().item()
    assert float((r1[0] - ref_rows[0]).abs().max()) <= (2 * TOL + 1e-5) * scale
    assert float((r1 - r0).abs().max() / r0.abs().max()) <= (2e-3 if small else 1e-5)


def test_recorded_backward_statistics_launch_keeps_its_kernel_family():
    """HR_OP_CONV i[17] (hrnet_conv_route): a plan sizes a backward-statistics launch's rows buffer for the kernel family
    chosen when it was recorded; flipping hrnet_conv_ring_enable() afterwards must not change how many rows the launch
    writes. Route 2 (ring) with the ring switched off fails loudly, route 1 (tile walk) with the ring on writes the
    tile-walking body's rows."""
    hh, C = _h(), _C()
    from hipnet._capi import HrOp
    N, H, W, Cc = 8, 16, 16, 128
    g = torch.Generator().manual_seed(3)
    dy = _q(torch.randn(N, Cc, H, W, generator=g))
    w = _q(torch.randn(Cc, Cc, 3, 3, generator=g) / np.sqrt(Cc * 9))
    bs_y = _q(torch.randn(N, Cc, H, W, generator=g))
    wp, _, _ = hh.pack_weights(w, DT, mode=1)
    dyd, byd = hh.nhwc(dy, DT), hh.nhwc(bs_y, DT)
    C.call('hrnet_conv_ring_enable', 1)
    assert C.call('hrnet_conv_route', 1, N, H, W, Cc, Cc, 3, 1) == 2
    rows_ring = C.call('hrnet_conv_rows_bwdstats', 1, N, H, W, Cc, Cc, 3, 1)
    C.call('hrnet_conv_ring_enable', 0)
    assert C.call('hrnet_conv_route', 1, N, H, W, Cc, Cc, 3, 1) == 1
    rows_walk = C.call('hrnet_conv_rows_bwdstats', 1, N, H, W, Cc, Cc, 3, 1)
    assert rows_ring != rows_walk, 'the case must distinguish the two families'

    def run(route, rows_n):
        y = torch.zeros(N, H, W, Cc, dtype=DT, device=hh.DEV)
        rows = torch.full((max(rows_ring, rows_walk) + 1, 2, Cc), float('nan'), dtype=torch.float32, device=hh.DEV)
        op = HrOp()
        op.kind = C.OP_CONV
        for k, val in enumerate((1, N, H, W, Cc, H, W, Cc, 3, 1, 0, 0, 0, 0, 0, 0, 0, route)):
            op.i[k] = val
        for k, t in ((0, dyd), (1, wp), (5, y), (6, rows), (7, byd)):
            op.p[k] = C.ptr(t)
        C.call('hrnet_program_run', ctypes.byref(op), 1, C.stream_ptr())
        hh.sync()
        written = int((~torch.isnan(rows[:, 0, 0])).sum())
        assert written == rows_n, (route, written, rows_n)
        return rows[:rows_n].double().sum(0).cpu()

    C.call('hrnet_conv_ring_enable', 1)
    r_walk = run(1, rows_walk)              # recorded for the tile walk: the ring switch does not pull it over
    r_ring = run(2, rows_ring)
    assert float((r_walk - r_ring).abs().max() / r_ring.abs().max()) <= 1e-5
    C.call('hrnet_conv_ring_enable', 0)
    with pytest.raises(RuntimeError, match='recorded for the LDS-ring'):
        run(2, rows_ring)
    run(1, rows_walk)
    C.call('hrnet_conv_ring_enable', 1)


SUM_CASES = [
    # N, H, W, Cin, Cout, BatchNorm given as
    (70, 64, 64, 32, 32, 'sums'),      # > 512 tiles: several tiles per workgroup, both register sets of the identity term
    (3, 20, 37, 32, 32, 'arrays'),     # tiles overhang the map on both axes
    (1, 16, 16, 32, 32, 'sums'),       # one tile per workgroup (a single stage)
    (40, 32, 32, 64, 64, 'sums'),      # two K chunks per tile, two output-channel blocks (the second must not write the sum)
    (2, 24, 50, 64, 64, 'arrays'),
    (5, 17, 16, 64, 32, 'sums'),       # an odd number of stages per workgroup
]


@pytest.mark.parametrize('case', SUM_CASES)
def test_ring_conv_with_the_residual_sum_matches_the_tile_walking_body(case):
    """hrnet_conv2d_sum on the LDS-ring pipeline (round 4): a = relu(bn(x) + x2) formed while a stage is transformed in
    LDS (the identity term in registers from issue to transform), written to `side` once per pixel, y = conv(a) with
    its batch statistics - against the tile-walking body (bit-identical `side` and y) and torch
    (pose_hrnet.py:54-55 + :44 of the next block)."""
    hh, C = _h(), _C()
    N, H, W, Cin, Cout, mode = case
    assert C.call('hrnet_conv_ring_enable', 1) is not None
    g = torch.Generator().manual_seed(3 + N + Cin + Cout + H)
    x = _q(torch.randn(N, Cin, H, W, generator=g) * 1.5 + 0.3)
    x2 = _q(torch.randn(N, Cin, H, W, generator=g))
    w = _q(torch.randn(Cout, Cin, 3, 3, generator=g) / np.sqrt(Cin * 9))
    gamma, beta = torch.rand(Cin, generator=g) + 0.5, torch.rand(Cin, generator=g) - 0.5
    cnt = float(N * H * W)
    s1, s2 = x.double().sum((0, 2, 3)), (x.double() ** 2).sum((0, 2, 3))
    frac = torch.rand(8, 1, generator=g).double()
    frac /= frac.sum()
    sums = torch.stack([frac * s1[None], frac * s2[None]], 1).float().contiguous()
    s1f, s2f = sums[:, 0].double().sum(0), sums[:, 1].double().sum(0)
    mean = s1f / cnt
    invstd = 1.0 / torch.sqrt((s2f / cnt - mean * mean).clamp_min(0).float() + 1e-5)
    scale = gamma * invstd
    shift = beta - mean.float() * scale
    a_ref = _q(F.relu(x * scale.view(1, -1, 1, 1) + shift.view(1, -1, 1, 1) + x2))
    y_ref = F.conv2d(a_ref, w, None, padding=1)
    wp, _, _ = hh.pack_weights(w, DT)
    xd, x2d = hh.nhwc(x, DT), hh.nhwc(x2, DT)
    gb = torch.cat([gamma, beta]).to(hh.DEV).contiguous()          # contiguous gamma | beta: what the ring form takes
    sums_d, scale_d, shift_d = sums.to(hh.DEV), scale.to(hh.DEV), shift.to(hh.DEV)
    outs = []
    prev_sum = C.call('hrnet_conv_ring_sum_enable', 1)      # (off by default: slower inside the training step)
    for ring in (1, 0):
        C.call('hrnet_conv_ring_enable', ring)
        side = torch.full((N, H, W, Cin), float('nan'), dtype=DT, device=hh.DEV)
        y = torch.full((N, H, W, Cout), float('nan'), dtype=DT, device=hh.DEV)
        st = torch.zeros(8, 2, Cout, dtype=torch.float32, device=hh.DEV)
        if mode == 'sums':
            C.call('hrnet_conv2d_sum', 1, xd.data_ptr(), x2d.data_ptr(), wp.data_ptr(), None, None, sums_d.data_ptr(),
                   gb.data_ptr(), gb.data_ptr() + 4 * Cin, 1.0 / cnt, 1e-5, side.data_ptr(), y.data_ptr(), st.data_ptr(), 1,
                   N, H, W, Cin, Cout, 3, C.stream_ptr())
        else:
            C.call('hrnet_conv2d_sum', 1, xd.data_ptr(), x2d.data_ptr(), wp.data_ptr(), scale_d.data_ptr(), shift_d.data_ptr(),
                   None, None, None, 0.0, 0.0, side.data_ptr(), y.data_ptr(), st.data_ptr(), 1, N, H, W, Cin, Cout, 3,
                   C.stream_ptr())
        hh.sync()
        outs.append((side, y, st.double().sum(0).cpu()))
    C.call('hrnet_conv_ring_enable', 1)
    (sd1, y1, st1), (sd0, y0, st0) = outs
    assert not torch.isnan(sd1.float()).any() and not torch.isnan(y1.float()).any()
    assert torch.equal(sd1.view(torch.int16), sd0.view(torch.int16)), 'the sum written out differs from the tile-walking body'
    assert torch.equal(y1.view(torch.int16), y0.view(torch.int16)), 'ring and tile-walking bodies differ'
    assert hh.rel_err(hh.from_nhwc(sd1, Cin), a_ref) <= 2.0 ** -7
    assert hh.rel_err(hh.from_nhwc(y1, Cout), y_ref) <= 2 * TOL + 1e-2
    assert float((st1 - st0).abs().max() / st0.abs().max()) <= 1e-5
    # the kernel really is the ring's (a name query of the same shape says so)
    import ctypes
    buf = ctypes.create_string_buffer(160)
    C.call('hrnet_conv_kernel_name', 1, N, H, W, Cin, Cout, 3, 1, 0, 5, buf, 160)
    assert buf.value.decode().startswith('conv_ring_kernel<') and buf.value.decode().endswith('false, true>'), buf.value
    C.call('hrnet_conv_ring_sum_enable', prev_sum)
    C.call('hrnet_conv_kernel_name', 1, N, H, W, Cin, Cout, 3, 1, 0, 5, buf, 160)
    assert prev_sum != 0 or buf.value.decode().startswith('conv_fwds_kernel<'), buf.value
