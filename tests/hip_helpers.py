"""Thin tensor-level wrappers over the C ABI for the GPU parity tests."""
import ctypes

import numpy as np
import torch

from hipnet import _capi as C

DEV = 'cuda:0'


def dt_id(dtype):
    return C.dtype_id(dtype)


def sync():
    torch.cuda.synchronize()


def nhwc(x_nchw, dtype, cpad=None):
    """CPU NCHW f32 -> device NHWC `dtype` (channels zero-padded to cpad)."""
    n, c, h, w = x_nchw.shape
    cp = cpad or c
    t = torch.zeros(n, h, w, cp, dtype=torch.float32)
    t[..., :c] = x_nchw.permute(0, 2, 3, 1)
    return t.to(dtype).to(DEV).contiguous()


def from_nhwc(t, c=None):
    """device NHWC -> CPU NCHW f32"""
    t = t.float().cpu()
    if c is not None:
        t = t[..., :c]
    return t.permute(0, 3, 1, 2).contiguous()


def pack_weights(w_oihw, dtype, mode=0, cout_pad=None, cin_pad=None):
    co, ci, ks, _ = w_oihw.shape
    cout_pad = cout_pad or (co + 15) // 16 * 16
    if mode == 2:
        cin_pad = cin_pad or 32
        n = cout_pad * cin_pad
    else:
        cin_pad = cin_pad or (ci + 7) // 8 * 8
        n = cout_pad * ks * ks * cin_pad
    wd = w_oihw.float().to(DEV).contiguous()
    out = torch.empty(n, dtype=dtype, device=DEV)
    C.call('hrnet_pack_weights', dt_id(dtype), wd.data_ptr(), out.data_ptr(), co, ci, ks, cout_pad, cin_pad, mode,
           C.stream_ptr())
    return out, cout_pad, cin_pad


def conv2d(x, w_packed, N, H, W, Cin, Cout, ks, stride, dtype, in_scale=None, in_shift=None, bias=None,
           in_relu=False, stats=False, upz=False, out=None, accumulate=False, out_hw=None):
    pad = ks // 2
    if upz:
        Ho, Wo = out_hw
    else:
        Ho, Wo = (H + 2 * pad - ks) // stride + 1, (W + 2 * pad - ks) // stride + 1
    y = out if out is not None else torch.empty(N, Ho, Wo, Cout, dtype=dtype, device=DEV)
    st = None
    if stats:
        tiles = C.call('hrnet_conv_tiles', N, Ho, Wo, Cout, ks, stride)
        st = torch.zeros(tiles, 2, Cout, dtype=torch.float32, device=DEV)
    C.call('hrnet_conv2d', dt_id(dtype), x.data_ptr(), w_packed.data_ptr(), C.ptr(in_scale), C.ptr(in_shift),
           C.ptr(bias), y.data_ptr(), C.ptr(st), N, H, W, Cin, Ho, Wo, Cout, ks, stride, 1 if upz else 0,
           1 if in_relu else 0, 1 if accumulate else 0, C.stream_ptr())
    return y, st


def wgrad(x, dy, N, H, W, Cin, Ho, Wo, Cout, ks, stride, dtype, in_scale=None, in_shift=None, in_relu=False,
          cout_real=None, cin_real=None):
    ns = C.call('hrnet_wgrad_splits', dt_id(dtype), N, Ho, Wo, Cout, Cin, ks, stride)
    slabs = torch.full((ns, Cout, ks * ks, Cin), float('nan'), dtype=torch.float32, device=DEV)
    C.call('hrnet_conv2d_wgrad', dt_id(dtype), x.data_ptr(), dy.data_ptr(), C.ptr(in_scale), C.ptr(in_shift),
           slabs.data_ptr(), N, H, W, Cin, Ho, Wo, Cout, ks, stride, 1 if in_relu else 0, ns, C.stream_ptr())
    cout_real = cout_real or Cout
    cin_real = cin_real or Cin
    g = torch.zeros(cout_real, cin_real, ks, ks, dtype=torch.float32, device=DEV)
    C.call('hrnet_wgrad_reduce', slabs.data_ptr(), g.data_ptr(), ns, Cout, Cin, ks, cout_real, cin_real, 0, 0,
           C.stream_ptr())
    return g


def ptr_array(tensors):
    arr = (ctypes.c_void_p * len(tensors))(*[C.ptr(t) for t in tensors])
    return arr


def int_array(vals):
    return (ctypes.c_int * len(vals))(*vals)


def rel_err(a, b):
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    return np.abs(a - b).max() / max(np.abs(b).max(), 1e-12)
