"""TEST INFRASTRUCTURE ONLY - CPU restatement of the reference's deformable convolution v1.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.

The reference op is CUDA-only (lib/deformable_conv/src/cpu/deform_cpu.cpp:24,45 raise "Not
implement on cpu"), so it cannot be run in this container or on the ROCm box. This file restates the
published algorithm from the reference's sources and is pinned by the invariants the reference's
own test.py checks (zero offsets == nn.Conv2d incl. groups, test.py:37-69; identity kernel,
test.py:113-141; im2col_step invariance, test.py:177-248; finite-difference gradcheck in float64
with eps=1e-3, atol=1e-3, rtol=1e-2, test.py:377-400), see tests/test_oracle_golden.py.

Two independent forms are given and checked against each other:
  * deform_conv_forward / deform_conv_backward: numpy, the explicit column / col2im / coordinate
    formulas of deform_im2col_cuda.cuh, vectorised over pixels;
  * deform_conv_torch: the forward written with differentiable torch ops, whose autograd gradients
    must agree with the explicit backward.
"""
import numpy as np


def _out_size(H, W, kh, kw, stride, padding, dilation):
    # deform_conv_cuda.cu:57-58
    Ho = (H + 2 * padding[0] - (dilation[0] * (kh - 1) + 1)) // stride[0] + 1
    Wo = (W + 2 * padding[1] - (dilation[1] * (kw - 1) + 1)) // stride[1] + 1
    return Ho, Wo


def _corners(h, w, H, W):
    """Corner indices, bilinear weights and validity (deform_im2col_cuda.cuh:24-53). h, w: arrays."""
    hl = np.floor(h).astype(np.int64)
    wl = np.floor(w).astype(np.int64)
    hh_, wh = hl + 1, wl + 1
    lh, lw = h - hl, w - wl
    hh, hw = 1.0 - lh, 1.0 - lw
    inside = (h > -1) & (w > -1) & (h < H) & (w < W)          # deform_im2col_cuda.cuh:173
    cs = [(hl, wl, hh * hw, (hl >= 0) & (wl >= 0)),
          (hl, wh, hh * lw, (hl >= 0) & (wh <= W - 1)),
          (hh_, wl, lh * hw, (hh_ <= H - 1) & (wl >= 0)),
          (hh_, wh, lh * lw, (hh_ <= H - 1) & (wh <= W - 1))]
    return cs, inside, (lh, lw, hh, hw)


def _sample_positions(offset, b_shape, kh, kw, stride, padding, dilation, DG):
    """h_im, w_im of every (b, dg, tap, y, x) (deform_im2col_cuda.cuh:160-172): the offset tensor
    holds, per deformable group and tap k = i*kw+j, the dy plane at channel 2k and dx at 2k+1."""
    B, Ho, Wo = b_shape
    K = kh * kw
    off = offset.reshape(B, DG, K, 2, Ho, Wo)
    i = (np.arange(K) // kw).reshape(1, 1, K, 1, 1)
    j = (np.arange(K) % kw).reshape(1, 1, K, 1, 1)
    y = np.arange(Ho).reshape(1, 1, 1, Ho, 1)
    x = np.arange(Wo).reshape(1, 1, 1, 1, Wo)
    h = y * stride[0] - padding[0] + i * dilation[0] + off[:, :, :, 0]
    w = x * stride[1] - padding[1] + j * dilation[1] + off[:, :, :, 1]
    return h, w                                               # [B,DG,K,Ho,Wo]


def deform_columns(input, offset, kh, kw, stride, padding, dilation, DG):
    """The column buffer of deformable_im2col: cols[b, c, k, y, x]."""
    B, C, H, W = input.shape
    Ho, Wo = _out_size(H, W, kh, kw, stride, padding, dilation)
    h, w = _sample_positions(offset, (B, Ho, Wo), kh, kw, stride, padding, dilation, DG)
    cpd = C // DG
    h = np.repeat(h, cpd, axis=1)                             # [B,C,K,Ho,Wo]
    w = np.repeat(w, cpd, axis=1)
    cs, inside, _ = _corners(h, w, H, W)
    bi = np.arange(B).reshape(B, 1, 1, 1, 1)
    ci = np.arange(C).reshape(1, C, 1, 1, 1)
    cols = np.zeros(h.shape, dtype=input.dtype)
    for (ih, iw, wt, ok) in cs:
        ok = ok & inside
        v = input[bi, ci, np.clip(ih, 0, H - 1), np.clip(iw, 0, W - 1)]
        cols += np.where(ok, wt * v, 0)
    return cols


def deform_conv_forward(input, offset, weight, bias, stride, padding, dilation, groups, DG):
    """deform_conv_cuda_forward (deform_conv_cuda.cu:19-136): out = bias + W (x) columns per group."""
    Co, Cg, kh, kw = weight.shape
    B, C, H, W = input.shape
    cols = deform_columns(input, offset, kh, kw, stride, padding, dilation, DG)
    Ho, Wo = cols.shape[-2:]
    Og = Co // groups
    out = np.zeros((B, Co, Ho, Wo), dtype=input.dtype)
    for g in range(groups):
        wg = weight[g * Og:(g + 1) * Og].reshape(Og, Cg * kh * kw)
        cg = cols[:, g * Cg:(g + 1) * Cg].reshape(B, Cg * kh * kw, Ho * Wo)
        out[:, g * Og:(g + 1) * Og] = np.einsum('ok,bkp->bop', wg, cg).reshape(B, Og, Ho, Wo)
    if bias is not None:
        out += bias.reshape(1, Co, 1, 1)
    return out


def deform_conv_backward(input, offset, weight, grad_output, stride, padding, dilation, groups, DG):
    """deform_conv_cuda_backward (deform_conv_cuda.cu:139-271): returns grad_input, grad_offset,
    grad_weight, grad_bias with the col2im (deform_im2col_cuda.cuh:192-246) and coordinate
    (deform_im2col_cuda.cuh:249-310, weights :82-124) formulas."""
    Co, Cg, kh, kw = weight.shape
    B, C, H, W = input.shape
    K = kh * kw
    Og = Co // groups
    cols = deform_columns(input, offset, kh, kw, stride, padding, dilation, DG)
    Ho, Wo = cols.shape[-2:]
    grad_weight = np.zeros_like(weight)
    gcols = np.zeros_like(cols)                               # d loss / d columns
    for g in range(groups):
        go = grad_output[:, g * Og:(g + 1) * Og].reshape(B, Og, Ho * Wo)
        cg = cols[:, g * Cg:(g + 1) * Cg].reshape(B, Cg * K, Ho * Wo)
        grad_weight[g * Og:(g + 1) * Og] = np.einsum('bop,bkp->ok', go, cg).reshape(Og, Cg, kh, kw)
        wg = weight[g * Og:(g + 1) * Og].reshape(Og, Cg * K)
        gcols[:, g * Cg:(g + 1) * Cg] = np.einsum('ok,bop->bkp', wg, go).reshape(B, Cg, K, Ho, Wo)
    grad_bias = grad_output.sum(axis=(0, 2, 3))

    h, w = _sample_positions(offset, (B, Ho, Wo), kh, kw, stride, padding, dilation, DG)
    cpd = C // DG
    hC, wC = np.repeat(h, cpd, axis=1), np.repeat(w, cpd, axis=1)
    cs, inside, (lh, lw, hh, hw) = _corners(hC, wC, H, W)
    bi = np.broadcast_to(np.arange(B).reshape(B, 1, 1, 1, 1), hC.shape)
    ci = np.broadcast_to(np.arange(C).reshape(1, C, 1, 1, 1), hC.shape)
    grad_input = np.zeros_like(input)
    vals = []
    for (ih, iw, wt, ok) in cs:
        ok = ok & inside
        ihc, iwc = np.clip(ih, 0, H - 1), np.clip(iw, 0, W - 1)
        np.add.at(grad_input, (bi[ok], ci[ok], ihc[ok], iwc[ok]), (gcols * wt)[ok])
        vals.append(np.where(ok, input[bi, ci, ihc, iwc], 0))
    v1, v2, v3, v4 = vals
    # d sample / d h and / d w; zero outside (the reference sets inv_h = inv_w = -2 there)
    dh = np.where(inside, -hw * v1 - lw * v2 + hw * v3 + lw * v4, 0)
    dw = np.where(inside, -hh * v1 + hh * v2 - lh * v3 + lh * v4, 0)
    gh = (gcols * dh).reshape(B, DG, cpd, K, Ho, Wo).sum(axis=2)
    gw = (gcols * dw).reshape(B, DG, cpd, K, Ho, Wo).sum(axis=2)
    grad_offset = np.stack([gh, gw], axis=3).reshape(offset.shape)
    return grad_input, grad_offset, grad_weight, grad_bias


def deform_conv_torch(input, offset, weight, bias, stride, padding, dilation, groups, DG, mask=None):
    """The forward in differentiable torch ops (any float dtype) - autograd provides a second,
    independent statement of the gradients. mask [B, DG*K, Ho, Wo]: the modulated form (DCNv2,
    modulated_deform_im2col_cuda.cuh:128-193: column = sample * mask)."""
    import torch
    B, C, H, W = input.shape
    Co, Cg, kh, kw = weight.shape
    K = kh * kw
    Ho, Wo = _out_size(H, W, kh, kw, stride, padding, dilation)
    off = offset.reshape(B, DG, K, 2, Ho, Wo)
    dt, dev = input.dtype, input.device
    i = (torch.arange(K, device=dev) // kw).reshape(1, 1, K, 1, 1).to(dt)
    j = (torch.arange(K, device=dev) % kw).reshape(1, 1, K, 1, 1).to(dt)
    y = torch.arange(Ho, device=dev).reshape(1, 1, 1, Ho, 1).to(dt)
    x = torch.arange(Wo, device=dev).reshape(1, 1, 1, 1, Wo).to(dt)
    cpd = C // DG
    h = (y * stride[0] - padding[0] + i * dilation[0] + off[:, :, :, 0]).repeat_interleave(cpd, dim=1)
    w = (x * stride[1] - padding[1] + j * dilation[1] + off[:, :, :, 1]).repeat_interleave(cpd, dim=1)
    inside = (h > -1) & (w > -1) & (h < H) & (w < W)
    hl, wl = torch.floor(h.detach()), torch.floor(w.detach())
    lh, lw = h - hl, w - wl
    hl, wl = hl.long(), wl.long()
    flat = input.reshape(B, C, 1, H * W).expand(B, C, K, H * W)
    cols = torch.zeros_like(h)
    for (ih, iw, wt) in ((hl, wl, (1 - lh) * (1 - lw)), (hl, wl + 1, (1 - lh) * lw),
                         (hl + 1, wl, lh * (1 - lw)), (hl + 1, wl + 1, lh * lw)):
        ok = inside & (ih >= 0) & (ih <= H - 1) & (iw >= 0) & (iw <= W - 1)
        idx = (ih.clamp(0, H - 1) * W + iw.clamp(0, W - 1)).reshape(B, C, K, Ho * Wo)
        v = torch.gather(flat, 3, idx).reshape(B, C, K, Ho, Wo)
        cols = cols + torch.where(ok, wt * v, torch.zeros_like(v))
    if mask is not None:
        cols = cols * mask.reshape(B, DG, K, Ho, Wo).repeat_interleave(cpd, dim=1)
    Og = Co // groups
    outs = []
    for g in range(groups):
        wg = weight[g * Og:(g + 1) * Og].reshape(Og, Cg * K)
        cg = cols[:, g * Cg:(g + 1) * Cg].reshape(B, Cg * K, Ho * Wo)
        outs.append(torch.einsum('ok,bkp->bop', wg, cg).reshape(B, Og, Ho, Wo))
    out = torch.cat(outs, dim=1)
    if bias is not None:
        out = out + bias.reshape(1, Co, 1, 1)
    return out


# ---- modulated deformable convolution (DCNv2) ------------------------------------------------------------------------
# Reference: lib/deformable_conv/src/cuda/modulated_deform_im2col_cuda.cuh:128-193 (column = bilinear sample * mask,
# mask channel = deformable group * K + tap), :196-257 (col2im: the scattered input gradient carries the mask),
# :259-330 (coordinate kernel: offset gradient carries the mask, grad_mask = sum over the group's channels of
# d loss / d column * unmasked sample); host modulated_deform_conv_cuda.cu:20-285 (shapes, bias, groups). CUDA-only
# like v1; pinned by the reference test.py's invariants for the modulated op (zero offsets + unit mask == nn.Conv2d,
# test.py:69-110; identity kernel 142-181; im2col_step 219-260, 304-349; gradcheck 405-434) in tests/test_dcn_oracle.py.
def _mask_per_channel(mask, B, C, DG, K, Ho, Wo):
    return np.repeat(mask.reshape(B, DG, K, Ho, Wo), C // DG, axis=1)                  # [B,C,K,Ho,Wo]


def modulated_deform_conv_forward(input, offset, mask, weight, bias, stride, padding, dilation, groups, DG):
    Co, Cg, kh, kw = weight.shape
    B, C, H, W = input.shape
    cols = deform_columns(input, offset, kh, kw, stride, padding, dilation, DG)
    Ho, Wo = cols.shape[-2:]
    cols = cols * _mask_per_channel(mask, B, C, DG, kh * kw, Ho, Wo)
    Og = Co // groups
    out = np.zeros((B, Co, Ho, Wo), dtype=input.dtype)
    for g in range(groups):
        wg = weight[g * Og:(g + 1) * Og].reshape(Og, Cg * kh * kw)
        cg = cols[:, g * Cg:(g + 1) * Cg].reshape(B, Cg * kh * kw, Ho * Wo)
        out[:, g * Og:(g + 1) * Og] = np.einsum('ok,bkp->bop', wg, cg).reshape(B, Og, Ho, Wo)
    if bias is not None:
        out += bias.reshape(1, Co, 1, 1)
    return out


def modulated_deform_conv_backward(input, offset, mask, weight, grad_output, stride, padding, dilation, groups, DG):
    """returns grad_input, grad_offset, grad_mask, grad_weight, grad_bias"""
    Co, Cg, kh, kw = weight.shape
    B, C, H, W = input.shape
    K = kh * kw
    Og = Co // groups
    cols_u = deform_columns(input, offset, kh, kw, stride, padding, dilation, DG)      # unmasked samples
    Ho, Wo = cols_u.shape[-2:]
    m = _mask_per_channel(mask, B, C, DG, K, Ho, Wo)
    cols = cols_u * m
    grad_weight = np.zeros_like(weight)
    gcols = np.zeros_like(cols)                               # d loss / d (masked) columns
    for g in range(groups):
        go = grad_output[:, g * Og:(g + 1) * Og].reshape(B, Og, Ho * Wo)
        cg = cols[:, g * Cg:(g + 1) * Cg].reshape(B, Cg * K, Ho * Wo)
        grad_weight[g * Og:(g + 1) * Og] = np.einsum('bop,bkp->ok', go, cg).reshape(Og, Cg, kh, kw)
        wg = weight[g * Og:(g + 1) * Og].reshape(Og, Cg * K)
        gcols[:, g * Cg:(g + 1) * Cg] = np.einsum('ok,bop->bkp', wg, go).reshape(B, Cg, K, Ho, Wo)
    grad_bias = grad_output.sum(axis=(0, 2, 3))
    cpd = C // DG
    grad_mask = (gcols * cols_u).reshape(B, DG, cpd, K, Ho, Wo).sum(axis=2).reshape(mask.shape)
    gs = gcols * m                                            # d loss / d sample
    h, w = _sample_positions(offset, (B, Ho, Wo), kh, kw, stride, padding, dilation, DG)
    hC, wC = np.repeat(h, cpd, axis=1), np.repeat(w, cpd, axis=1)
    cs, inside, (lh, lw, hh, hw) = _corners(hC, wC, H, W)
    bi = np.broadcast_to(np.arange(B).reshape(B, 1, 1, 1, 1), hC.shape)
    ci = np.broadcast_to(np.arange(C).reshape(1, C, 1, 1, 1), hC.shape)
    grad_input = np.zeros_like(input)
    vals = []
    for (ih, iw, wt, ok) in cs:
        ok = ok & inside
        ihc, iwc = np.clip(ih, 0, H - 1), np.clip(iw, 0, W - 1)
        np.add.at(grad_input, (bi[ok], ci[ok], ihc[ok], iwc[ok]), (gs * wt)[ok])
        vals.append(np.where(ok, input[bi, ci, ihc, iwc], 0))
    v1, v2, v3, v4 = vals
    dh = np.where(inside, -hw * v1 - lw * v2 + hw * v3 + lw * v4, 0)
    dw = np.where(inside, -hh * v1 + hh * v2 - lh * v3 + lh * v4, 0)
    gh = (gs * dh).reshape(B, DG, cpd, K, Ho, Wo).sum(axis=2)
    gw = (gs * dw).reshape(B, DG, cpd, K, Ho, Wo).sum(axis=2)
    grad_offset = np.stack([gh, gw], axis=3).reshape(offset.shape)
    return grad_input, grad_offset, grad_mask, grad_weight, grad_bias
