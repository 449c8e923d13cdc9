#!/usr/bin/env python3
"""Where a wave's cycles go in the generic direct weight-gradient kernel on the Discriminator's stride-2 layers (s_memtime stamps).

Needs the attribution build (conv_wgrad.hip compiled with -DWG_TIME=1):
    FCD_LIB=build_exp/libfcdgan_wgtime.so python tools/wgrad_segments.py [--md gpurun_out/r04_wgrad_segments.md]"""
import argparse
import ctypes
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fcd_gan_pytorch_amd import _ops as ops          # noqa: E402
from fcd_gan_pytorch_amd._lib import lib, check, LIB_PATH      # noqa: E402

SEG = ['prologue (first tile\'s LDS-DMA + barrier)', 'issue of the next tile\'s LDS-DMA (29 wave-instructions per workgroup)',
       'MFMA loop of the tile (72 MFMAs 32x32x2 per wave)', 'tile barrier (incl. the wait for the next tile\'s DMA)', 'epilogue (partial dW store)']


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--md', default=None)
    args = ap.parse_args()
    raw = ctypes.CDLL(LIB_PATH)
    if not hasattr(raw, 'fcd_wgrad_time_buf'):
        raise SystemExit('not the attribution build: FCD_LIB=build_exp/libfcdgan_wgtime.so')
    raw.fcd_wgrad_time_buf.argtypes = [ctypes.c_void_p]
    s = ops._stream()
    tbuf = torch.zeros(1 << 21, dtype=torch.int64, device='cuda')
    L = ['# Generic direct weight-gradient kernel on the stride-2 layers: where a wave\'s cycles go (tools/wgrad_segments.py)', '',
         'Attribution build (-DWG_TIME=1).  4 waves per workgroup (64 k x 64 c x 9 taps), two workgroups resident per CU, one tile = 16 output positions.', '']
    for tag, N, C, HW, K, ST in (('G / S 64->64 @256 (N=16), rolling 3x3 kernel', 16, 64, 256, 64, 1), ('D 64->128 s2 @128 (N=32)', 32, 64, 128, 128, 2), ('D 128->256 s2 @64 (N=32)', 32, 128, 64, 256, 2),
                             ('D 256->512 s2 @32 (N=32)', 32, 256, 32, 512, 2)):
        x = torch.randn(N, C, HW, HW, device='cuda')
        w = torch.randn(K, C, 3, 3, device='cuda') * 0.05
        d = ops._desc(x.shape, w.shape, ST, 1)
        dy = torch.randn(N, K, d.P, d.Q, device='cuda')
        dw = torch.empty_like(w)
        nb = lib.fcd_conv2d_bwd_weight_ws_bytes(ctypes.byref(d))
        ws = torch.empty(max(nb, 16), dtype=torch.uint8, device='cuda')

        def run():
            check(lib.fcd_conv2d_bwd_weight(ctypes.byref(d), ops._p(x), ops._p(dy), None, ops._p(dw), ops._p(ws), ws.numel(), s))
        raw.fcd_wgrad_time_buf(ctypes.c_void_p(0))
        run(); run()
        torch.cuda.synchronize()
        tbuf.zero_()
        raw.fcd_wgrad_time_buf(ctypes.c_void_p(tbuf.data_ptr()))
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); run(); e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1)
        t = tbuf.cpu().numpy().reshape(-1, 4, 8).astype(np.float64)
        t = t[t[:, :, 6].sum(axis=1) > 0]
        wgs = t.shape[0]
        tot = t[:, :, 6]
        seg = t[:, :, :5]
        other = tot - seg.sum(axis=2)
        tick_us = 1.0 / 2100.0          # ~2.1 GHz shader clock (the whole call also runs two re-layout passes: no launch-time calibration)
        print('\n%s: %d workgroups, whole call %.3f ms, mean wave life %.1f us (at 2.1 GHz)' % (tag, wgs, ms, tot.mean() * tick_us))
        L += ['## %s' % tag, '', '%d workgroups (%.1f per resident slot); whole call (re-layout + kernel + reduce) %.3f ms; mean wave life %.1f us at 2.1 GHz'
              % (wgs, wgs / 512.0, ms, tot.mean() * tick_us), '', '| segment | share of the wave\'s life | mean us per wave |', '|---|---|---|']
        for i, name in enumerate(SEG):
            v = seg[:, :, i]
            print('  %-70s %5.1f %%  %8.2f us' % (name[:70], 100 * v.sum() / tot.sum(), v.mean() * tick_us))
            L.append('| %s | %.1f %% | %.2f |' % (name, 100 * v.sum() / tot.sum(), v.mean() * tick_us))
        print('  %-70s %5.1f %%' % ('(between the stamps)', 100 * other.sum() / tot.sum()))
        L += ['| (between the stamps) | %.1f %% | |' % (100 * other.sum() / tot.sum()), '']
    if args.md:
        os.makedirs(os.path.dirname(os.path.abspath(args.md)), exist_ok=True)
        open(args.md, 'w').write('\n'.join(L) + '\n')


if __name__ == '__main__':
    main()
