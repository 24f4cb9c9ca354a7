"""CPU tests of the data steps either side of the hot path:
* the 2-D evaluator's accumulators / result files (core/evaluate2d.py, reference tools/evaluate_2D.py:165-294)
  against the line-by-line restatement in oracle/eval2d_cpu.py and the format of a result pair the reference
  committed (tests/golden/ref_eval2D_*: data files copied from /root/reference/tools/eval2D_results_*);
* the host mirror of the Gaussian target generator (hipnet.synth.gaussian_heatmaps) against targets.npz, which
  tests/golden/make_golden_r2.py produced by running the reference's own HeatmapGenerator."""
import os

import numpy as np
import pytest

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')


def _batches(rng, n, B, K, rhd):
    out = []
    for _ in range(n):
        gt = rng.uniform(0, 64, (B, K, 2)).astype(np.float32)
        pred = gt + rng.normal(0, 3.0, (B, K, 2)).astype(np.float32)
        vis = (rng.uniform(size=(B, K, 1)) < 0.8).astype(np.float32)
        b = dict(pred=pred, gt=gt, visibility=vis)
        if rhd:
            b['crop_size'] = rng.uniform(80, 300, B).astype(np.float32)
            b['corner'] = rng.uniform(0, 100, (B, 2)).astype(np.float32)
        else:
            b['orig_size'] = (640, 480)
        out.append(b)
    return out


@pytest.mark.parametrize('rhd', [True, False])
def test_accumulator_matches_the_restated_reference_loop(rhd, tmp_path):
    from core.evaluate2d import Eval2DAccumulator
    from oracle import eval2d_cpu as O
    rng = np.random.default_rng(5)
    K = 21
    batches = _batches(rng, 5, 7, K, rhd)
    batches[2]['visibility'][:] = 0            # an all-invisible batch
    batches[0]['pred'][0, 0] = batches[0]['gt'][0, 0] + np.array([3.0, 4.0])    # error exactly 5 px * scale
    acc = Eval2DAccumulator(K, 64)
    for b in batches:
        kw = dict(crop_size=b['crop_size'], corner=b['corner']) if rhd else dict(orig_size=b['orig_size'])
        acc.add(b['pred'], b['gt'], b['visibility'], **kw)
    mse, pck = acc.save(str(tmp_path))
    ref_mse, ref_pck = O.evaluate_batches(batches, K, 64)
    np.testing.assert_allclose(mse, ref_mse, rtol=1e-12)
    np.testing.assert_array_equal(pck[0], ref_pck[0])
    np.testing.assert_allclose(pck[1], ref_pck[1], rtol=1e-12)
    # files parse like the pair the reference committed
    ref_dir = os.path.join(GOLD, 'ref_eval2D_RHD_HRNet_w32_max_hmloss_v1')
    for name in ('PCK2d.txt', 'mse2d_each_joint.txt'):
        ours, theirs = np.loadtxt(os.path.join(str(tmp_path), name)), np.loadtxt(os.path.join(ref_dir, name))
        assert ours.shape == theirs.shape
        line_o = open(os.path.join(str(tmp_path), name)).readline()
        line_t = open(os.path.join(ref_dir, name)).readline()
        assert len(line_o.split()) == len(line_t.split())
        # same number format: '%.4f' for the error file, numpy's default '%.18e' for the PCK file
        assert line_o.split()[0][::-1].find('.') == line_t.split()[0][::-1].find('.') or 'e' in line_t
        assert ('e' in line_o) == ('e' in line_t)
    theirs = np.loadtxt(os.path.join(ref_dir, 'PCK2d.txt'))
    np.testing.assert_array_equal(theirs[0], np.arange(1, 50))
    assert np.all(np.diff(theirs[1]) >= 0) and np.all(np.diff(pck[1]) >= 0)
    # the figures SURVEY section 6 quotes from this file pair: PCK@20px 0.9415, mean EPE 5.77 px
    assert abs(theirs[1, 19] - 0.9415) < 5e-4
    assert abs(np.loadtxt(os.path.join(ref_dir, 'mse2d_each_joint.txt')).mean() - 5.77) < 5e-3


def test_strict_threshold_and_invisible_joints():
    from core.evaluate2d import Eval2DAccumulator
    acc = Eval2DAccumulator(2, 64)
    gt = np.zeros((1, 2, 2))
    pred = np.array([[[5.0, 0.0], [100.0, 0.0]]])
    acc.add(pred, gt, np.array([[1.0, 0.0]]), orig_size=(64, 64))
    mse, pck = acc.result()
    assert mse[0] == 5.0 and np.isnan(mse[1])          # never-visible joint: 0/0 as in the reference
    assert pck[1, 4] == 0.0 and pck[1, 5] == 1.0       # strict '<': an error of exactly 5 px misses threshold 5


def test_host_gaussian_targets_match_the_reference_generator():
    from hipnet import synth
    g = np.load(os.path.join(GOLD, 'targets.npz'))
    for case, res in (('A', 64), ('B', 128)):
        j = g[case + '.joints']
        got = synth.gaussian_heatmaps(j[..., :2], j[..., 2:3] > 0, res, res, sigma=int(g[case + '.sigma']))
        ref = g[case + '.heatmaps']
        assert got.shape == ref.shape
        assert np.array_equal(got > 0, ref > 0)                 # identical windows, skips and clipping
        np.testing.assert_allclose(got, ref, rtol=0, atol=2e-7)  # f32 exp vs the reference's f64 exp rounded once
    # the corner / edge cases placed by hand in case A really are in the fixture
    ref = g['A.heatmaps']
    assert ref[0, 0, 0, 0] == 1.0 and ref[0, 1, 63, 63] == 1.0 and ref[0, 5].max() == 0.0 and ref[0, 9].max() == 0.0
    assert ref[0, 6, 5, 0] == 1.0        # x = -0.5 truncates to 0 (int()), still inside the map
    assert ref[0, 7].max() == 0.0        # x = -1.0 is outside
