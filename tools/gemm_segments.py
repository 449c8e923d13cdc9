#!/usr/bin/env python3
"""Where a wave's cycles go in the 256-tile split GEMM of the Winograd F(4x4,3x3) path (s_memtime stamps inside the kernel).

Needs the attribution build of the library (conv_wino.hip compiled with -DYG_TIME=1, see tools/build_gemmtime.sh):
    FCD_LIB=build_exp/libfcdgan_gemmtime.so python tools/gemm_segments.py [--md gpurun_out/r04_gemm_segments.md]
Runs the forward Winograd call of the perception VGG's conv3_x / conv4_x / conv3_1 layers (208 band images) -- their batched GEMMs are
the four biggest launches of the step -- reads the per-wave cycle sums and prints, per launch, the share of a wave's life in each
segment of its stage loop."""
import argparse
import ctypes
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fcd_gan_pytorch_amd import _ops as ops          # noqa: E402
from fcd_gan_pytorch_amd._lib import lib, check, LIB_PATH      # noqa: E402

SEG = ['stage prologue: first LDS reads (B raw, A fragments of group 0) + bf16 split of step 0',
       'four MFMA groups (96 MFMAs) with the split of step 1, the A-fragment reads and the LDS-DMA issue of the next stage in between',
       'stage barrier (behind the wave\'s own DMA wait: skew between the eight waves + the slowest wave\'s DMA)',
       'C store of a finished transform position (per batch, not per stage)',
       's_waitcnt vmcnt(0): the wave\'s own share of the next stage\'s LDS-DMA (issued behind the second MFMA group)']


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--md', default=None)
    ap.add_argument('--n', type=int, default=208)
    args = ap.parse_args()
    raw = ctypes.CDLL(LIB_PATH)
    if not hasattr(raw, 'fcd_wino_gemm_time_buf'):
        raise SystemExit('this is not the attribution build: FCD_LIB=build_exp/libfcdgan_gemmtime.so (tools/build_gemmtime.sh)')
    raw.fcd_wino_gemm_time_buf.argtypes = [ctypes.c_void_p]
    N = args.n
    s = ops._stream()
    tbuf = torch.zeros(1 << 21, dtype=torch.int64, device='cuda')         # [workgroup][8 waves][8]
    L = ['# 256-tile split GEMM: where a wave\'s cycles go (s_memtime stamps, tools/gemm_segments.py)', '',
         'Attribution build (-DYG_TIME=1: the stamps are scheduling fences and wait for the wave\'s outstanding LDS reads; launch times are 20 - 25 % above',
         'the product build: read the SHARES, not the microseconds).  Forward Winograd calls of the perception VGG on %d band images; one workgroup (8 waves, 256 x 256 tile) resident per CU; a' % N,
         'stage = 32 reduction elements = 96 MFMAs per wave.  One counter tick is calibrated against the launch time (workgroups run back to back on a CU).', '']
    for tag, C, HW, K in (('conv3_x  256 -> 256 @ 64 x 64', 256, 64, 256), ('conv4_x  512 -> 512 @ 32 x 32', 512, 32, 512),
                          ('conv3_1  128 -> 256 @ 64 x 64', 128, 64, 256)):
        x = torch.randn(N, C, HW, HW, device='cuda')
        w = torch.randn(K, C, 3, 3, device='cuda') * 0.05
        b = torch.zeros(K, device='cuda')
        d = ops._desc(x.shape, w.shape, 1, 1)
        m = lib.fcd_conv_wino_plan(ctypes.byref(d), 0)
        y = torch.empty(N, K, HW, HW, device='cuda')
        U = ops.wino_weight(w, 0, m)
        ws = torch.empty(lib.fcd_conv_wino_ws_bytes(ctypes.byref(d), 0), dtype=torch.uint8, device='cuda')

        def run():
            check(lib.fcd_conv2d_fwd_wino(ctypes.byref(d), ops._p(x), ops._p(U), ops._p(b), ops._p(y), 0, None, None,
                                          ops._p(ws), ws.numel(), s))
        raw.fcd_wino_gemm_time_buf(ctypes.c_void_p(0))
        run(); run()
        torch.cuda.synchronize()
        from fcd_gan_pytorch_amd import _lib
        _lib.prof_read(reset=True)
        lib.fcd_prof_enable(2)
        tbuf.zero_()
        raw.fcd_wino_gemm_time_buf(ctypes.c_void_p(tbuf.data_ptr()))
        run()
        torch.cuda.synchronize()
        lib.fcd_prof_enable(0)
        _lib.prof_read(reset=True)
        det = _lib.prof_detail(reset=True)
        gm = [e for e in det if e['family'] in ('wino_gemm', 'wino_gemm_bf16x6')]
        gemm_ms = sum(e['ms'] for e in gm) if gm else None
        t = tbuf.cpu().numpy().reshape(-1, 8, 8).astype(np.float64)
        used = t[:, :, 6].sum(axis=1) > 0
        t = t[used]
        wgs = t.shape[0]
        tot = t[:, :, 6]
        per_cu = -(-wgs // 256)
        seg = t[:, :, :5]
        other = tot - seg.sum(axis=2)
        # tick: the sum of a CU's workgroup lives ~= the launch; without a per-kernel time fall back to 1 / 2.0 GHz
        tick_us = (1e3 * gemm_ms / (tot.mean() * wgs / 256.0)) if gemm_ms else 0.5e-3
        print('\n%s: %d workgroups (%d per CU), GEMM launch %s ms, mean wave life %.1f us' % (
            tag, wgs, per_cu, ('%.3f' % gemm_ms) if gemm_ms else '?', tot.mean() * tick_us))
        L += ['## %s' % tag, '', '%d workgroups (%.1f per CU); GEMM launch %s ms; mean wave life %.1f us' % (
            wgs, wgs / 256.0, ('%.3f' % gemm_ms) if gemm_ms else '(not timed; tick taken as 0.5 ns)', tot.mean() * tick_us), '',
            '| segment | mean us per wave | share of the wave\'s life | slowest wave of the workgroup, mean us |', '|---|---|---|---|']
        for i, name in enumerate(SEG):
            v = seg[:, :, i]
            print('  %-70s %8.2f us  %5.1f %%  (max over the 8 waves: %.2f us)' % (name[:70], v.mean() * tick_us, 100 * v.sum() / tot.sum(),
                                                                                 v.max(axis=1).mean() * tick_us))
            L.append('| %s | %.2f | %.1f %% | %.2f |' % (name, v.mean() * tick_us, 100 * v.sum() / tot.sum(), v.max(axis=1).mean() * tick_us))
        print('  %-70s %8.2f us  %5.1f %%' % ('(before the first stage / between the stamps)', other.mean() * tick_us, 100 * other.sum() / tot.sum()))
        L += ['| (before the first stage, between the stamps) | %.2f | %.1f %% | |' % (other.mean() * tick_us, 100 * other.sum() / tot.sum()), '']
        del x, y, ws
    if args.md:
        os.makedirs(os.path.dirname(os.path.abspath(args.md)), exist_ok=True)
        with open(args.md, 'w') as f:
            f.write('\n'.join(L) + '\n')


if __name__ == '__main__':
    main()
