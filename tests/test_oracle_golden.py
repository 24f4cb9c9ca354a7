"""Pin the CPU oracle (oracle/hrnet_cpu.py) to fixtures produced by the reference's own
module (tests/golden/make_golden.py). CPU only."""
import os

import numpy as np
import pytest
import torch

from hipnet import synth
from oracle import hrnet_cpu as O


def _state(salt=0, overrides=None):
    tmpl = O.state_template()
    sd = {k: torch.from_numpy(np.asarray(v)) for k, v in synth.fill_state_dict(tmpl, salt).items()}
    for k, v in (overrides or {}).items():
        sd[k] = torch.from_numpy(v)
    return sd


def _checksum(t):
    a = t.detach().double().reshape(-1)
    n = a.numel()
    idx = (np.arange(16, dtype=np.int64) * 2654435761 % n)
    return np.concatenate([[a.sum().item(), a.abs().sum().item()], a[idx].numpy()])


def test_state_template_matches_reference_inventory():
    tmpl = O.state_template()
    assert len(tmpl) == 1839                       # 921 params + 918 BN buffers
    n_params = sum(int(np.prod(s)) for k, s in tmpl.items()
                   if not k.endswith(('running_mean', 'running_var', 'num_batches_tracked')))
    assert n_params == 29547477


def test_eval_forward_matches_reference(golden_dir):
    g = np.load(os.path.join(golden_dir, 'w32_eval_b1.npz'))
    stats = {k[5:]: g[k] for k in g.files if k.startswith('stat.')}
    sd = _state(0, stats)
    x = torch.from_numpy(synth.rhd_batch(1, seed=1234)['imgs'])
    with torch.no_grad():
        hm, inter, _ = O.hrnet_forward(sd, O.W32_EXTRA, x, training=False)
    assert np.abs(hm.numpy() - g['heatmaps']).max() <= 1e-5 * max(1.0, np.abs(g['heatmaps']).max())
    np.testing.assert_allclose(_checksum(inter), g['inter_feat_checksum'], rtol=1e-5, atol=1e-5)
    np.testing.assert_allclose(inter[0, :, 10, 7:23].numpy(), g['inter_feat_slice'], rtol=1e-4, atol=1e-5)


def test_train_forward_backward_matches_reference(golden_dir):
    g = np.load(os.path.join(golden_dir, 'w32_train_b4.npz'))
    sd = _state(0)
    keys = [str(k) for k in g['grad_keys']]
    for k in keys:
        sd[k].requires_grad_(True)
    b = synth.rhd_batch(4, seed=1234)
    hm, inter, new_stats = O.hrnet_forward(sd, O.W32_EXTRA, torch.from_numpy(b['imgs']), training=True)
    loss = O.heatmap_loss(hm, torch.from_numpy(b['heatmaps']))
    loss.backward()
    assert abs(loss.item() - float(g['heatmap_loss'])) <= 1e-5 * abs(float(g['heatmap_loss']))
    scale = max(1.0, np.abs(g['heatmaps0']).max())
    assert np.abs(hm[0].detach().numpy() - g['heatmaps0']).max() <= 2e-5 * scale
    cs = np.array([[sd[k].grad.double().sum().item(), sd[k].grad.double().abs().sum().item()] for k in keys])
    np.testing.assert_allclose(cs[:, 1], g['grad_checksums'][:, 1], rtol=2e-4)
    for k in g.files:
        if k.startswith('grad.'):
            ref = g[k]
            got = sd[k[5:]].grad.numpy()
            assert np.abs(got - ref).max() <= 2e-4 * max(np.abs(ref).max(), 1e-6), k
        if k.startswith('stat.'):
            np.testing.assert_allclose(new_stats[k[5:]].numpy(), g[k], rtol=1e-4, atol=1e-6)


def test_small_train_matches_reference(golden_dir):
    g = np.load(os.path.join(golden_dir, 'w32_small_train_b2.npz'))
    sd = _state(3)
    b = synth.rhd_batch(2, seed=99, img_h=64, img_w=64)
    hm, inter, _ = O.hrnet_forward(sd, O.W32_EXTRA, torch.from_numpy(b['imgs']), training=True)
    assert np.abs(hm.detach().numpy() - g['heatmaps']).max() <= 2e-5 * max(1.0, np.abs(g['heatmaps']).max())
    assert np.abs(inter.detach().numpy() - g['inter_feat']).max() <= 2e-5 * max(1.0, np.abs(g['inter_feat']).max())


def test_losses_and_decode_match_reference(golden_dir):
    g = np.load(os.path.join(golden_dir, 'micro.npz'))
    p, t = torch.from_numpy(g['hl_pred']), torch.from_numpy(g['hl_gt'])
    assert abs(O.heatmap_loss(p, t, 'l2').item() - float(g['hl_l2'])) <= 1e-5 * float(g['hl_l2'])
    assert abs(O.heatmap_loss(p, t, 'l1').item() - float(g['hl_l1'])) <= 1e-5 * float(g['hl_l1'])
    pp, pg, vis = (torch.from_numpy(g[k]) for k in ('jm_pred', 'jm_gt', 'jm_vis'))
    assert abs(O.joints_mse_loss(pp, pg, vis).item() - float(g['jm_vis_loss'])) <= 1e-5 * float(g['jm_vis_loss'])
    assert abs(O.joints_mse_loss(pp, pg).item() - float(g['jm_novis_loss'])) <= 1e-5 * float(g['jm_novis_loss'])
    assert O.joints_mse_loss(pp, pg, torch.zeros(5, 21)).item() == float(g['jm_allinvis_loss']) == 0.0
    pred = O.get_final_preds(torch.from_numpy(g['am_hm']), use_softmax=False)
    assert np.array_equal(pred.numpy(), g['am_pred'])


def test_argmax_decode_uses_height_as_row_stride_like_reference():
    # heatmap_decoding.py:103-106 uses shape[2] (H) for both % and //, also on non-square maps
    hm = torch.zeros(1, 1, 4, 6)
    hm[0, 0, 2, 5] = 1.0                      # flat index 17
    pred = O.get_final_preds(hm, use_softmax=False)
    assert pred[0, 0].tolist() == [17 % 4, 17 // 4]


def test_expectation_decode_definition():
    hm = torch.zeros(1, 2, 8, 8)
    hm[0, 0, 3, 5] = 1.0
    hm[0, 1, 1, 2] = 0.5
    hm[0, 1, 7, 6] = 0.5
    pred = O.get_final_preds(hm, use_softmax=True)
    assert torch.allclose(pred[0, 0], torch.tensor([5.0, 3.0]))
    assert torch.allclose(pred[0, 1], torch.tensor([4.0, 4.0]))


def test_get_max_preds_mask():
    hm = np.zeros((1, 2, 4, 5), dtype=np.float32)
    hm[0, 0, 3, 1] = 2.0
    hm[0, 1] = -1.0
    preds, maxvals = O.get_max_preds(hm)
    assert preds[0, 0].tolist() == [1.0, 3.0] and preds[0, 1].tolist() == [0.0, 0.0]
    assert maxvals[0, 0, 0] == 2.0 and maxvals[0, 1, 0] == -1.0


def test_softmax_head_variant_matches_reference_fixture(golden_dir):
    """pose_hrnet_softmax: soft-max heat maps, loss, temperature gradient and every
    parameter gradient of the reference's own module (tests/golden/make_golden_softmax.py)."""
    g = np.load(os.path.join(golden_dir, 'w32_softmax_train_b2.npz'))
    sd = _state(5)
    sd['trainable_temp'] = torch.tensor(1.5)
    pk = [k for k in sd if not k.endswith(('running_mean', 'running_var', 'num_batches_tracked'))]
    for k in pk:
        sd[k] = sd[k].double().requires_grad_(True)
    for k in sd:
        if k.endswith(('running_mean', 'running_var')):
            sd[k] = sd[k].double()
    b = synth.rhd_batch(2, seed=321, img_h=128, img_w=128)
    hm, inter, _ = O.hrnet_forward(sd, O.W32_EXTRA, torch.from_numpy(b['imgs']).double(), training=True,
                                   softmax_head=True)
    gt = torch.from_numpy(b['heatmaps']).double()
    gt = gt / gt.sum((2, 3), keepdim=True).clamp_min(1e-6)
    loss = O.heatmap_loss(hm, gt) * 1e4 + 1e-3 * inter.square().mean()
    loss.backward()
    assert tuple(inter.shape) == tuple(g['inter_shape']) == (2, 480, 32, 32)
    assert np.abs(hm.detach().numpy() - g['heatmaps']).max() <= 2e-5      # fp64 here, fp32 fixture; maps sum to 1
    assert abs(loss.item() - float(g['loss'])) <= 1e-4 * abs(float(g['loss']))
    assert abs(sd['trainable_temp'].grad.item() - float(g['temp_grad'])) <= 2e-3 * abs(float(g['temp_grad']))
    np.testing.assert_allclose(_checksum(inter)[:2], g['inter_feat_checksum'][:2], rtol=1e-4)
    for k in ('last_layer.3.weight', 'stage4.2.fuse_layers.0.3.0.weight', 'conv1.weight'):
        ref = g['grad.' + k]
        err = np.abs(sd[k].grad.numpy() - ref).max() / max(np.abs(ref).max(), 1e-12)
        assert err <= 5e-2, (k, err)          # the fixture is fp32: the chaotic-gradient band
    assert list(g['state_keys_head'])[0] == 'trainable_temp'


def test_expectation_decode_cross_checked_by_the_reference_centre_of_mass(golden_dir):
    """kornia (the reference's decode) is unavailable; the reference's own integrate_tensor_2d
    (triangulation_model_utils/op.py:11-47) gives the same numbers on normalised maps and, with softmax=True,
    pins the softmax-head + expectation chain of pose_hrnet_softmax."""
    g = np.load(os.path.join(golden_dir, 'decode_crosscheck.npz'))
    got = O.get_final_preds(torch.from_numpy(g['pos']), use_softmax=True).numpy()
    assert np.abs(got - g['coords_normalised']).max() <= 2e-5
    soft = torch.softmax(torch.from_numpy(g['raw']).reshape(3, 21, -1), dim=2).reshape(3, 21, 24, 20)
    assert np.abs(soft.numpy() - g['softmax_maps']).max() <= 1e-7
    got2 = O.get_final_preds(soft, use_softmax=True).numpy()
    assert np.abs(got2 - g['coords_softmax']).max() <= 2e-5


def _w48_extra():
    extra = dict(O.W32_EXTRA)
    for s_, ch in ((2, [48, 96]), (3, [48, 96, 192]), (4, [48, 96, 192, 384])):
        extra['STAGE{}'.format(s_)] = dict(O.W32_EXTRA['STAGE{}'.format(s_)], NUM_CHANNELS=ch)
    return extra


def test_w48_eval_forward_matches_reference_fixture(golden_dir):
    """BASELINE config 4 geometry (48/96/192/384 channels, 384x288): the oracle against the reference's own
    pose_hrnet and pose_hrnet_softmax modules (tests/golden/make_golden_w48.py)."""
    g = np.load(os.path.join(golden_dir, 'w48_eval_b1.npz'))
    extra = _w48_extra()
    tmpl = O.state_template(extra)
    assert sum(int(np.prod(v)) for k, v in tmpl.items()
               if not k.endswith(('running_mean', 'running_var', 'num_batches_tracked'))) == int(g['n_params'])
    sd = {k: torch.from_numpy(np.asarray(v)) for k, v in synth.fill_state_dict(tmpl, 6).items()}
    for k in g.files:
        if k.startswith('stat.'):
            sd[k[5:]] = torch.from_numpy(g[k])
    x = torch.from_numpy(synth.rhd_batch(1, seed=2, img_h=384, img_w=288)['imgs'])
    with torch.no_grad():
        hm, _, _ = O.hrnet_forward(sd, extra, x, training=False)
        assert tuple(hm.shape) == tuple(g['plain.shape']) == (1, 21, 96, 72)
        np.testing.assert_allclose(hm[0, :, 40, 20:44].numpy(), g['plain.heatmaps_slice'], rtol=2e-4, atol=2e-4)
        np.testing.assert_allclose(_checksum(hm)[:2], g['plain.heatmaps_checksum'][:2], rtol=2e-4)
        # (the generator filled the temperature from the same keyed PRNG as every other entry)
        sd['trainable_temp'] = torch.from_numpy(np.asarray(synth.fill_state_dict({'trainable_temp': ()}, 6)['trainable_temp']))
        sd['last_layer.1.running_mean'] = torch.from_numpy(g['softmax.stat.last_layer.1.running_mean'])
        sd['last_layer.1.running_var'] = torch.from_numpy(g['softmax.stat.last_layer.1.running_var'])
        hs, _, _ = O.hrnet_forward(sd, extra, x, training=False, softmax_head=True)
        np.testing.assert_allclose(hs[0, :, 40, 20:44].numpy(), g['softmax.heatmaps_slice'], rtol=1e-3, atol=1e-8)


def test_core_inference_oracle_matches_reference_fixture(golden_dir):
    """the oracle's get_max_preds / final_preds_oracle against outputs of the reference's own
    lib/core/inference.py functions (tests/golden/make_golden_inference.py; cv2.getAffineTransform is the one
    call restated there). Integer work: bit-exact; the image-space map: float64 solve vs the reference's
    float64 matrix applied to float32 coordinates -> 1e-4 of a pixel."""
    g = np.load(os.path.join(golden_dir, 'inference_preds.npz'))
    for tag in ('sq', 'rect'):
        preds, maxvals = O.get_max_preds(g['hm_' + tag])
        assert np.array_equal(preds, g['preds_' + tag]) and preds.dtype == g['preds_' + tag].dtype
        assert np.array_equal(maxvals, g['maxvals_' + tag])
    for pp in (0, 1):
        want, wmax = g['fp_preds_pp%d' % pp], g['fp_maxvals_pp%d' % pp]
        got, gmax = O.final_preds_oracle(bool(pp), g['fp_hm'], g['fp_center'], g['fp_scale'])
        assert np.array_equal(gmax, wmax)
        assert np.abs(got - want).max() <= 1e-4, np.abs(got - want).max()
    # the refinement moved something (the fixture exercises it) and left border / flat peaks alone
    d = g['fp_preds_pp1'] - g['fp_preds_pp0']
    assert np.abs(d).max() > 0 and np.all(d[:, :3] == 0) and np.all(d[2, 5] == 0)
