#!/usr/bin/env python3
"""Cycle-level attribution of the fused F(2x2,3x3) kernel from s_memtime stamps inside the kernel.

Needs the attribution build of the library (conv_wino2.hip compiled with -DW2_TIME=1, see tools/build_w2time.sh):
    FCD_LIB=build_exp/libfcdgan_w2time.so python tools/w2_segments.py [--md gpurun_out/r04_w2_segments.md]
Runs the two dominant launches of the step -- VGG conv1_2 forward (+ReLU + max-pool epilogue) and its pooled data gradient,
N = 208 band images, 64 -> 64 channels, 256 x 256 -- plus the plain forward, reads the per-wave cycle sums and prints, per
launch, the mean share of a wave's life spent in each segment of its stage loop."""
import argparse
import ctypes
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fcd_gan_pytorch_amd import _ops as ops          # noqa: E402
from fcd_gan_pytorch_amd._lib import lib, check, LIB_PATH      # noqa: E402

SEG = ['prologue (first filter + patch fetch, LDS commit, barrier)', 'issue: filter LDS-DMA + patch loads of the chunks ahead',
       'operand LDS reads + transforms + MFMA issue', 'LDS commit of the next patch (waits for its global loads)', 'stage barrier',
       'epilogue (output transform + stores)']


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--md', default=None)
    ap.add_argument('--n', type=int, default=208)
    args = ap.parse_args()
    raw = ctypes.CDLL(LIB_PATH)
    if not hasattr(raw, 'fcd_wino2_time_buf'):
        raise SystemExit('this is not the attribution build: FCD_LIB=build_exp/libfcdgan_w2time.so (tools/build_w2time.sh)')
    raw.fcd_wino2_time_buf.argtypes = [ctypes.c_void_p]
    N, C, HW, K = args.n, 64, 256, 64
    s = ops._stream()
    x = torch.randn(N, C, HW, HW, device='cuda').relu_()
    w = torch.randn(K, C, 3, 3, device='cuda') * 0.05
    b = torch.zeros(K, device='cuda')
    d = ops._desc(x.shape, w.shape, 1, 1)
    y = torch.empty(N, K, HW, HW, device='cuda')
    py = torch.empty(N, K, HW // 2, HW // 2, device='cuda')
    code = torch.empty(N, K, HW // 2, HW // 2, dtype=torch.uint8, device='cuda')
    dpy = torch.randn_like(py)
    dx = torch.empty_like(x)
    U0, U1 = ops.wino2_weight(w, 0), ops.wino2_weight(w, 1)
    wgs = N * (HW // 8) * (HW // 32)
    tbuf = torch.zeros(wgs * 8 * 8, dtype=torch.int64, device='cuda')
    runs = [
        ('conv1_2 forward + ReLU + max-pool (EPI = 1)', lambda: check(lib.fcd_conv2d_fwd_wino2(
            ctypes.byref(d), ops._p(x), ops._p(U0), ops._p(b), None, 1, None, 0.0, None, ops._p(py), ops._p(code), s))),
        ('conv1_2 data gradient, pooled source (SRC = 2)', lambda: check(lib.fcd_conv2d_bwd_data_wino2(
            ctypes.byref(d), ops._p(dpy), None, ops._p(code), ops._p(U1), ops._p(dx), s))),
        ('64 -> 64 forward, plain', lambda: check(lib.fcd_conv2d_fwd_wino2(
            ctypes.byref(d), ops._p(x), ops._p(U0), ops._p(b), ops._p(y), 1, None, 0.0, None, None, None, s))),
    ]
    L = ['# Fused F(2x2,3x3) kernel: where a wave\'s cycles go (s_memtime stamps, tools/w2_segments.py)', '',
         'Attribution build (-DW2_TIME=1; the stamps wait for the wave\'s outstanding LDS reads: launch times are a few % above the product build).',
         'VGG conv1_2 geometry: N = %d band images, 64 -> 64 channels, 256 x 256; %d workgroups of 8 waves on 256 CUs (one resident workgroup per CU), '
         '8 stages of 8 channels per workgroup.  One counter tick is calibrated against the launch time (workgroups run back to back on a CU); it comes out at the '
         'shader clock (~0.44 ns).' % (N, wgs), '']
    for tag, fn in runs:
        raw.fcd_wino2_time_buf(ctypes.c_void_p(0))
        fn(); fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        tbuf.zero_()
        raw.fcd_wino2_time_buf(ctypes.c_void_p(tbuf.data_ptr()))
        e0.record(); fn(); e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1)
        t = tbuf.cpu().numpy().reshape(wgs, 8, 8).astype(np.float64)
        tot = t[:, :, 6]
        # the counter runs at the shader clock here (not at a fixed reference): calibrate one tick against the launch time --
        # a CU runs its wgs / 256 workgroups back to back, so launch time ~= mean wave life x workgroups per CU
        tick_us = 1e3 * ms / (tot.mean() * wgs / 256.0)
        seg = t[:, :, :6]
        other = tot - seg.sum(axis=2)
        print('\n%s: launch %.3f ms, mean wave life %.2f us, timer tick %.4f us' % (tag, ms, tot.mean() * tick_us, tick_us))
        L += ['## %s' % tag, '', 'launch %.3f ms (HIP events); mean wave life %.2f us; workgroup lives back to back on a CU: %.3f ms (x %d per CU)'
              % (ms, tot.mean() * tick_us, tot.mean() * tick_us * wgs / 256 / 1e3, wgs // 256), '',
              '| segment | mean us per wave | share of the wave\'s life | slowest wave of the workgroup, mean us |', '|---|---|---|---|']
        for i, name in enumerate(SEG):
            v = seg[:, :, i]
            print('  %-62s %8.2f us  %5.1f %%   (max over the 8 waves: %.2f us)' % (name, v.mean() * tick_us, 100 * v.sum() / tot.sum(),
                                                                                 v.max(axis=1).mean() * tick_us))
            L.append('| %s | %.2f | %.1f %% | %.2f |' % (name, v.mean() * tick_us, 100 * v.sum() / tot.sum(), v.max(axis=1).mean() * tick_us))
        print('  %-62s %8.2f us  %5.1f %%' % ('(between the stamps)', other.mean() * tick_us, 100 * other.sum() / tot.sum()))
        L += ['| (between the stamps) | %.2f | %.1f %% | |' % (other.mean() * tick_us, 100 * other.sum() / tot.sum()), '']
    if args.md:
        os.makedirs(os.path.dirname(os.path.abspath(args.md)), exist_ok=True)
        with open(args.md, 'w') as f:
            f.write('\n'.join(L) + '\n')


if __name__ == '__main__':
    main()
