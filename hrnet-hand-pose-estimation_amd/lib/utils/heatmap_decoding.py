"""Heat-map -> key-point decode (reference lib/utils/heatmap_decoding.py:87-107) on HIP kernels."""
import torch

from hipnet import _capi as C


class _ExpectationFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, hms):
        b, k, h, w = hms.shape
        preds = torch.empty((b, k, 2), dtype=torch.float32, device=hms.device)
        C.call('hrnet_decode_expectation', hms.data_ptr(), preds.data_ptr(), b * k, h, w, C.stream_ptr())
        ctx.shape = (b, k, h, w)
        return preds

    @staticmethod
    def backward(ctx, g):
        b, k, h, w = ctx.shape
        d = torch.empty((b, k, h, w), dtype=torch.float32, device=g.device)
        g = g.contiguous().float()
        C.call('hrnet_decode_expectation_bwd', g.data_ptr(), d.data_ptr(), b * k, h, w, 0, C.stream_ptr())
        return d


def get_final_preds(hms, use_softmax=True):
    """hms: B x K x H x W on the HIP device -> B x K x 2 [u right, v down].

    use_softmax=True : (sum x*h, sum y*h) over pixel coordinates (kornia spatial_expectation2d with
                       normalized_coordinates=False: no softmax inside); differentiable.
    use_softmax=False: first arg-max of the flattened map, u = idx % H, v = idx // H (H for both,
                       as the reference does), as float.
    """
    assert isinstance(hms, torch.Tensor), 'hms should be torch.Tensor'
    assert hms.ndim == 4, 'Heatmap shape should be 4-ndim'
    if not hms.is_cuda:
        raise RuntimeError('get_final_preds: expected a HIP-device tensor (no CPU path in this build)')
    hms = hms.contiguous().float()
    if use_softmax:
        return _ExpectationFn.apply(hms)
    b, k, h, w = hms.shape
    preds = torch.empty((b, k, 2), dtype=torch.float32, device=hms.device)
    C.call('hrnet_decode_argmax', hms.detach().data_ptr(), preds.data_ptr(), None, b * k, h, w, 0, C.stream_ptr())
    return preds
