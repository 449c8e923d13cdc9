#!/usr/bin/env python3
"""Where a wave's cycles go in the register-staged implicit-GEMM kernel on the sub-pixel data gradient of the Discriminator's
stride-2 layers (s_memtime stamps).  Needs the attribution build (conv_igemm.hip compiled with -DIG_TIME=1):
    FCD_LIB=build_exp/libfcdgan_igtime.so python tools/igemm_segments.py [--md gpurun_out/r04_igemm_segments.md]"""
import argparse
import ctypes
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fcd_gan_pytorch_amd import _ops as ops          # noqa: E402
from fcd_gan_pytorch_amd._lib import lib, check, LIB_PATH      # noqa: E402

SEG = ['prologue (first step\'s loads, LDS commit, barrier)', 'global loads of the next step issued (patch + filter slab into registers)',
       'MFMA block of the step (LDS operand reads + 32x32x2 MFMAs)', 'first barrier (everyone done reading the step\'s LDS image)',
       'LDS commit of the next step (waits for its global loads)', 'second barrier', 'epilogue (sub-pixel scatter of the four phases)']


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--md', default=None)
    args = ap.parse_args()
    raw = ctypes.CDLL(LIB_PATH)
    if not hasattr(raw, 'fcd_igemm_time_buf'):
        raise SystemExit('not the attribution build: FCD_LIB=build_exp/libfcdgan_igtime.so')
    raw.fcd_igemm_time_buf.argtypes = [ctypes.c_void_p]
    s = ops._stream()
    tbuf = torch.zeros(1 << 22, dtype=torch.int64, device='cuda')
    L = ['# Register-staged implicit-GEMM kernel on the sub-pixel stride-2 data gradients: where a wave\'s cycles go (tools/igemm_segments.py)', '',
         'Attribution build (-DIG_TIME=1).  4 waves per workgroup; a step = 8 reduction channels x 2 x 2 pseudo-taps (zero taps of a phase skipped).', '']
    for tag, N, C, HW, K in (('D 64->128 s2 @128 (N=32)', 32, 64, 128, 128), ('D 128->256 s2 @64 (N=32)', 32, 128, 64, 256),
                             ('D 256->512 s2 @32 (N=32)', 32, 256, 32, 512)):
        x = torch.randn(N, C, HW, HW, device='cuda')
        w = torch.randn(K, C, 3, 3, device='cuda') * 0.05
        d = ops._desc(x.shape, w.shape, 2, 1)
        dy = torch.randn(N, K, d.P, d.Q, device='cuda')
        dx = torch.empty_like(x)
        wp = torch.empty(lib.fcd_conv_s2_dgrad_packed_elems(K, C), dtype=torch.float32, device='cuda')
        check(lib.fcd_conv_s2_dgrad_pack(ops._p(w), ops._p(wp), K, C, s))

        def run():
            check(lib.fcd_conv2d_bwd_data_s2(ctypes.byref(d), ops._p(dy), None, ops._p(wp), ops._p(dx), s))
        raw.fcd_igemm_time_buf(ctypes.c_void_p(0))
        run(); run()
        torch.cuda.synchronize()
        tbuf.zero_()
        raw.fcd_igemm_time_buf(ctypes.c_void_p(tbuf.data_ptr()))
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); run(); e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1)
        t = tbuf.cpu().numpy().reshape(-1, 4, 8).astype(np.float64)
        t = t[t[:, :, 7].sum(axis=1) > 0]
        wgs = t.shape[0]
        tot = t[:, :, 7]
        seg = t[:, :, :7]
        other = tot - seg.sum(axis=2)
        flops = 2.0 * N * K * d.P * d.Q * C * 9
        print('\n%s: %d workgroups, launch %.3f ms (%.1f TFLOP/s algorithmic)' % (tag, wgs, ms, flops / ms / 1e9))
        L += ['## %s' % tag, '', '%d workgroups; launch %.3f ms = %.1f TFLOP/s on the algorithmic count' % (wgs, ms, flops / ms / 1e9), '',
              '| segment | share of the wave\'s life |', '|---|---|']
        for i, name in enumerate(SEG):
            v = seg[:, :, i]
            print('  %-75s %5.1f %%' % (name[:75], 100 * v.sum() / tot.sum()))
            L.append('| %s | %.1f %% |' % (name, 100 * v.sum() / tot.sum()))
        print('  %-75s %5.1f %%' % ('(between the stamps)', 100 * other.sum() / tot.sum()))
        L += ['| (between the stamps) | %.1f %% |' % (100 * other.sum() / tot.sum()), '']
    if args.md:
        os.makedirs(os.path.dirname(os.path.abspath(args.md)), exist_ok=True)
        open(args.md, 'w').write('\n'.join(L) + '\n')


if __name__ == '__main__':
    main()
