#!/usr/bin/env python3
"""Fused F(2x2,3x3) kernel vs the direct MFMA kernel on the 64-row layers of the headline workload (C-ABI calls)."""
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fcd_gan_pytorch_amd import _ops as ops          # noqa: E402
from fcd_gan_pytorch_amd._lib import lib, check       # noqa: E402

SHAPES = [  # tag, N, C, HW, K
    ('vgg conv1_2 64->64 @256 (N=208)', 208, 64, 256, 64),
    ('vgg conv2_1 64->128 @128 dgrad (N=208)', 208, 64, 128, 128),
    ('G 64->64 @256 (N=8)', 8, 64, 256, 64),
    ('S inc 64->64 @256 (N=16)', 16, 64, 256, 64),
]


def timeit(fn, iters=5):
    fn(); fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


def main():
    s = ops._stream()
    for tag, N, C, HW, K in (SHAPES[:1] if os.environ.get('W2_ONLY') else SHAPES):
        x = torch.randn(N, C, HW, HW, device='cuda').relu_()
        if os.environ.get('W2_ZERO'):      # all-zero activations: how much of the kernel time is the chip's power management?
            x.zero_()
        w = torch.randn(K, C, 3, 3, device='cuda') * 0.05
        b = torch.zeros(K, device='cuda')
        d = ops._desc(x.shape, w.shape, 1, 1)
        y = torch.empty(N, K, HW, HW, device='cuda')
        dy = torch.zeros_like(y) if os.environ.get('W2_ZERO') else torch.randn_like(y)
        dx = torch.empty_like(x)
        fl = 2.0 * N * K * HW * HW * C * 9
        line = '%-40s' % tag
        if lib.fcd_conv_wino2_plan(ctypes.byref(d), 0):
            U = ops.wino2_weight(w, 0)
            wp = ops.packed_weight(w, 0)
            t2 = timeit(lambda: check(lib.fcd_conv2d_fwd_wino2(ctypes.byref(d), ops._p(x), ops._p(U), ops._p(b), ops._p(y), 1, None, 0.0,
                                                               None, None, None, s)))
            y2 = y.clone()
            t1 = timeit(lambda: check(lib.fcd_conv2d_fwd(ctypes.byref(d), ops._p(x), ops._p(wp), ops._p(b), ops._p(y), 1, s)))
            err = (y2 - y).abs().max().item() / max(y.abs().max().item(), 1e-30)
            line += ' fwd: fused %.3f ms (%.0f TF alg) direct %.3f ms (%.0f TF)  rel diff %.1e |' % (t2, fl / t2 / 1e9, t1, fl / t1 / 1e9, err)
        if lib.fcd_conv_wino2_plan(ctypes.byref(d), 1):
            U = ops.wino2_weight(w, 1)
            wp = ops.packed_weight(w, 1)
            t2 = timeit(lambda: check(lib.fcd_conv2d_bwd_data_wino2(ctypes.byref(d), ops._p(dy), ops._p(y), None, ops._p(U), ops._p(dx), s)))
            d2 = dx.clone()
            t1 = timeit(lambda: check(lib.fcd_conv2d_bwd_data(ctypes.byref(d), ops._p(dy), ops._p(y), ops._p(wp), ops._p(dx), s)))
            err = (d2 - dx).abs().max().item() / max(dx.abs().max().item(), 1e-30)
            line += ' gated dgrad: fused %.3f ms (%.0f TF alg) direct %.3f ms (%.0f TF)  rel diff %.1e' % (t2, fl / t2 / 1e9, t1, fl / t1 / 1e9, err)
        if lib.fcd_conv_wino2_plan(ctypes.byref(d), 1) and lib.fcd_conv_wino2_plan(ctypes.byref(d), 0) and HW % 2 == 0:
            # the pooled data gradient (VGG conv1_2: gradient arrives on the max-pooled map + argmax codes)
            py = torch.empty(N, K, HW // 2, HW // 2, device='cuda')
            code = torch.empty(N, K, HW // 2, HW // 2, dtype=torch.uint8, device='cuda')
            check(lib.fcd_conv2d_fwd_wino2(ctypes.byref(d), ops._p(x), ops._p(ops.wino2_weight(w, 0)), ops._p(b), None, 1, None, 0.0, None,
                                           ops._p(py), ops._p(code), s))
            dpy = torch.randn_like(py)
            U = ops.wino2_weight(w, 1)
            t3 = timeit(lambda: check(lib.fcd_conv2d_bwd_data_wino2(ctypes.byref(d), ops._p(dpy), None, ops._p(code), ops._p(U), ops._p(dx), s)))
            line += ' | pooled dgrad: fused %.3f ms (%.0f TF alg)' % (t3, fl / t3 / 1e9)
        print(line)
        del x, y, dy, dx


if __name__ == '__main__':
    main()
