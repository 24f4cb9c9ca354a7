"""TEST INFRASTRUCTURE ONLY - the reference's three deformable-convolution device loops restated element by
element (pure Python, tiny cases only), index arithmetic included, so that the vectorised oracle
(oracle/dcn_cpu.py) is pinned to the reference's own indexing and boundary rules:

  columns()      deformable_im2col_gpu_kernel          deform_im2col_cuda.cuh:127-189 (+ bilinear :24-53)
  col2im()       deformable_col2im_gpu_kernel          :191-243 (+ dmcn_get_gradient_weight :56-80)
  col2im_coord() deformable_col2im_coord_gpu_kernel    :246-310 (+ dmcn_get_coordinate_weight :83-124)

The column buffer layout is the reference's: [c*kh*kw + i*kw + j][b][h_out][w_out].
"""
import math

import numpy as np


def _bilinear(im, height, width, h, w):
    h_low, w_low = math.floor(h), math.floor(w)
    h_high, w_high = h_low + 1, w_low + 1
    lh, lw = h - h_low, w - w_low
    hh, hw = 1 - lh, 1 - lw
    v1 = im[h_low, w_low] if (h_low >= 0 and w_low >= 0) else 0.0
    v2 = im[h_low, w_high] if (h_low >= 0 and w_high <= width - 1) else 0.0
    v3 = im[h_high, w_low] if (h_high <= height - 1 and w_low >= 0) else 0.0
    v4 = im[h_high, w_high] if (h_high <= height - 1 and w_high <= width - 1) else 0.0
    return hh * hw * v1 + hh * lw * v2 + lh * hw * v3 + lh * lw * v4


def _gradient_weight(ah, aw, h, w, height, width):
    if ah <= -1 or ah >= height or aw <= -1 or aw >= width:
        return 0.0
    hl, wl = math.floor(ah), math.floor(aw)
    hh, wh = hl + 1, wl + 1
    weight = 0.0
    if h == hl and w == wl:
        weight = (h + 1 - ah) * (w + 1 - aw)
    if h == hl and w == wh:
        weight = (h + 1 - ah) * (aw + 1 - w)
    if h == hh and w == wl:
        weight = (ah + 1 - h) * (w + 1 - aw)
    if h == hh and w == wh:
        weight = (ah + 1 - h) * (aw + 1 - w)
    return weight


def _coordinate_weight(ah, aw, height, width, im, bp_dir):
    if ah <= -1 or ah >= height or aw <= -1 or aw >= width:
        return 0.0
    hl, wl = math.floor(ah), math.floor(aw)
    hh, wh = hl + 1, wl + 1
    weight = 0.0
    if bp_dir == 0:
        if hl >= 0 and wl >= 0:
            weight += -1 * (wl + 1 - aw) * im[hl, wl]
        if hl >= 0 and wh <= width - 1:
            weight += -1 * (aw - wl) * im[hl, wh]
        if hh <= height - 1 and wl >= 0:
            weight += (wl + 1 - aw) * im[hh, wl]
        if hh <= height - 1 and wh <= width - 1:
            weight += (aw - wl) * im[hh, wh]
    else:
        if hl >= 0 and wl >= 0:
            weight += -1 * (hl + 1 - ah) * im[hl, wl]
        if hl >= 0 and wh <= width - 1:
            weight += (hl + 1 - ah) * im[hl, wh]
        if hh <= height - 1 and wl >= 0:
            weight += -1 * (ah - hl) * im[hh, wl]
        if hh <= height - 1 and wh <= width - 1:
            weight += (ah - hl) * im[hh, wh]
    return weight


def _geom(im_shape, kh, kw, stride, pad, dil):
    B, C, H, W = im_shape
    Ho = (H + 2 * pad[0] - (dil[0] * (kh - 1) + 1)) // stride[0] + 1
    Wo = (W + 2 * pad[1] - (dil[1] * (kw - 1) + 1)) // stride[1] + 1
    return B, C, H, W, Ho, Wo


def columns(im, offset, kh, kw, stride, pad, dil, DG):
    B, C, H, W, Ho, Wo = _geom(im.shape, kh, kw, stride, pad, dil)
    cpd = C // DG
    col = np.zeros((C * kh * kw, B, Ho, Wo))
    for c_im in range(C):
        for b in range(B):
            off = offset[b].reshape(DG, 2 * kh * kw, Ho, Wo)[c_im // cpd]
            for h_col in range(Ho):
                for w_col in range(Wo):
                    h_in, w_in = h_col * stride[0] - pad[0], w_col * stride[1] - pad[1]
                    for i in range(kh):
                        for j in range(kw):
                            h_im = h_in + i * dil[0] + off[2 * (i * kw + j), h_col, w_col]
                            w_im = w_in + j * dil[1] + off[2 * (i * kw + j) + 1, h_col, w_col]
                            val = 0.0
                            if h_im > -1 and w_im > -1 and h_im < H and w_im < W:
                                val = _bilinear(im[b, c_im], H, W, h_im, w_im)
                            col[c_im * kh * kw + i * kw + j, b, h_col, w_col] = val
    return col


def col2im(col_grad, offset, im_shape, kh, kw, stride, pad, dil, DG):
    B, C, H, W, Ho, Wo = _geom(im_shape, kh, kw, stride, pad, dil)
    cpd = C // DG
    grad_im = np.zeros(im_shape)
    for c in range(C):
        for i in range(kh):
            for j in range(kw):
                for b in range(B):
                    off = offset[b].reshape(DG, 2 * kh * kw, Ho, Wo)[c // cpd]
                    for h_out in range(Ho):
                        for w_out in range(Wo):
                            inv_h = h_out * stride[0] - pad[0] + i * dil[0] + off[2 * (i * kw + j), h_out, w_out]
                            inv_w = w_out * stride[1] - pad[1] + j * dil[1] + off[2 * (i * kw + j) + 1, h_out, w_out]
                            top = col_grad[c * kh * kw + i * kw + j, b, h_out, w_out]
                            cur_h, cur_w = int(inv_h), int(inv_w)           # C cast: truncation
                            for dy in range(-2, 3):
                                for dx in range(-2, 3):
                                    y, x = cur_h + dy, cur_w + dx
                                    if 0 <= y < H and 0 <= x < W and abs(inv_h - y) < 1 and abs(inv_w - x) < 1:
                                        grad_im[b, c, y, x] += _gradient_weight(inv_h, inv_w, y, x, H, W) * top
    return grad_im


def col2im_coord(col_grad, im, offset, kh, kw, stride, pad, dil, DG):
    B, C, H, W, Ho, Wo = _geom(im.shape, kh, kw, stride, pad, dil)
    cpd = C // DG
    K = kh * kw
    grad_offset = np.zeros(offset.shape)
    for b in range(B):
        for c in range(DG * 2 * K):                      # offset channel
            dg = c // (2 * K)
            offset_c = c - dg * 2 * K
            off = offset[b].reshape(DG, 2 * K, Ho, Wo)[dg]
            for h in range(Ho):
                for w in range(Wo):
                    val = 0.0
                    cnt = 0
                    # col_c walks the column rows of this group that belong to tap offset_c // 2
                    for col_c in range(offset_c // 2, cpd * K, K):
                        row = dg * cpd * K + col_c
                        j = row % kw
                        i = (row // kw) % kh
                        inv_h = h * stride[0] - pad[0] + i * dil[0] + off[2 * (i * kw + j), h, w]
                        inv_w = w * stride[1] - pad[1] + j * dil[1] + off[2 * (i * kw + j) + 1, h, w]
                        if inv_h <= -1 or inv_w <= -1 or inv_h >= H or inv_w >= W:
                            inv_h = inv_w = -2
                        weight = _coordinate_weight(inv_h, inv_w, H, W, im[b, dg * cpd + cnt], offset_c % 2)
                        val += weight * col_grad[row, b, h, w]
                        cnt += 1
                    grad_offset[b, c, h, w] = val
    return grad_offset
