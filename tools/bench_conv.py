#!/usr/bin/env python3
"""Per-layer micro-benchmark of the conv kernels on the shapes of the headline workload
(RSSS step, 13 bands, 256x256, 8 tile pairs per GPU).  Prints TFLOP/s for forward,
data-gradient and weight-gradient of every distinct layer shape."""
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fcd_gan_pytorch_amd import _ops as ops          # noqa: E402
from fcd_gan_pytorch_amd._lib import ConvDesc, lib, check  # noqa: E402

NB = int(os.environ.get('NB', '8'))
C0 = 13
SHAPES = []
# (tag, N, C, H, K, R, stride, pad, needs_wgrad)
vggN = 2 * NB * C0
cin = 3
hw = 256
for v in (64, 64, 'M', 128, 128, 'M', 256, 256, 256, 'M', 512, 512, 512, 'M', 512, 512, 512):
    if v == 'M':
        hw //= 2
        continue
    SHAPES.append(('vgg %d->%d @%d' % (cin, v, hw), vggN, cin, hw, v, 3, 1, 1, False))
    cin = v
enc = [(C0, 64, 256), (64, 64, 256), (64, 128, 128), (128, 128, 128), (128, 256, 64), (256, 256, 64),
       (256, 512, 32), (512, 512, 32), (512, 512, 16), (512, 512, 16)]
for ci, co, s in enc:
    SHAPES.append(('S enc %d->%d @%d' % (ci, co, s), 2 * NB, ci, s, co, 3, 1, 1, True))
dec = [(2048, 1024, 32), (1024, 512, 32), (1024, 512, 64), (512, 256, 64), (512, 256, 128), (256, 128, 128),
       (256, 128, 256), (128, 128, 256)]
for ci, co, s in dec:
    SHAPES.append(('S dec %d->%d @%d' % (ci, co, s), NB, ci, s, co, 3, 1, 1, True))
SHAPES.append(('S outc 128->1 @256', NB, 128, 256, 1, 1, 1, 0, True))
SHAPES.append(('G 13->64 9x9 @256', NB, C0, 256, 64, 9, 1, 4, False))
SHAPES.append(('G 64->64 @256', NB, 64, 256, 64, 3, 1, 1, False))
SHAPES.append(('G 64->13 9x9 @256', NB, 64, 256, C0, 9, 1, 4, False))
for ci, co, s in ((C0, 64, 256), (64, 128, 128), (128, 256, 64), (256, 512, 32)):
    SHAPES.append(('D %d->%d s2 @%d' % (ci, co, s), 4 * NB, ci, s, co, 3, 2, 1, True))


def timeit(fn, iters=5):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


def main():
    only = sys.argv[1] if len(sys.argv) > 1 else ''
    print('%-26s %5s %10s | %8s %8s %8s  (TFLOP/s; ms)' % ('layer', 'N', 'GFLOP', 'fwd', 'dgrad', 'wgrad'))
    tot = {'fwd': 0.0, 'dgrad': 0.0, 'wgrad': 0.0}
    for tag, N, C, HW, K, R, st, pad, wg in SHAPES:
        if only and only not in tag:
            continue
        x = torch.randn(N, C, HW, HW, device='cuda')
        w = torch.randn(K, C, R, R, device='cuda') * 0.05
        b = torch.zeros(K, device='cuda')
        d = ops._desc(x.shape, w.shape, st, pad)
        y = torch.empty(N, K, d.P, d.Q, device='cuda')
        dy = torch.randn_like(y)
        dx = torch.empty_like(x)
        dw = torch.empty_like(w)
        wp, wpb = ops.packed_weight(w, 0), ops.packed_weight(w, 1)
        s = ops._stream()
        flops = 2.0 * N * K * d.P * d.Q * C * R * R
        t_f = timeit(lambda: check(lib.fcd_conv2d_fwd(ctypes.byref(d), ops._p(x), ops._p(wp), ops._p(b), ops._p(y), 0, s)))
        t_d = timeit(lambda: check(lib.fcd_conv2d_bwd_data(ctypes.byref(d), ops._p(dy), None, ops._p(wpb), ops._p(dx), s)))
        res = '%8.1f %8.1f' % (flops / t_f / 1e9, flops / t_d / 1e9)
        ms = '%6.2f %6.2f' % (t_f, t_d)
        mw = lib.fcd_conv_wino_plan(ctypes.byref(d), 0) if st == 1 else 0
        md = lib.fcd_conv_wino_plan(ctypes.byref(d), 1) if st == 1 else 0
        wino = ''
        if mw:
            U = ops.wino_weight(w, 0, mw)
            wsb = torch.empty(lib.fcd_conv_wino_ws_bytes(ctypes.byref(d), 0), dtype=torch.uint8, device='cuda')
            t_fw = timeit(lambda: check(lib.fcd_conv2d_fwd_wino(ctypes.byref(d), ops._p(x), ops._p(U), ops._p(b), ops._p(y), 0,
                                                                None, None, ops._p(wsb), wsb.numel(), s)))
            wino += '  wino%d fwd %6.2f ms (%5.1f TF-eq)' % (mw, t_fw, flops / t_fw / 1e9)
            tot['fwd_best'] = tot.get('fwd_best', 0.0) + min(t_fw, t_f) - t_f
            del wsb
        if md:
            U1 = ops.wino_weight(w, 1, md)
            wsb = torch.empty(lib.fcd_conv_wino_ws_bytes(ctypes.byref(d), 1), dtype=torch.uint8, device='cuda')
            t_dw = timeit(lambda: check(lib.fcd_conv2d_bwd_data_wino(ctypes.byref(d), ops._p(dy), None, None, ops._p(U1),
                                                                     ops._p(dx), ops._p(wsb), wsb.numel(), s)))
            wino += '  dgrad %6.2f ms (%5.1f TF-eq)' % (t_dw, flops / t_dw / 1e9)
            tot['dgrad_best'] = tot.get('dgrad_best', 0.0) + min(t_dw, t_d) - t_d
            del wsb
        tot['fwd'] += t_f; tot['dgrad'] += t_d
        if wg:
            nb = lib.fcd_conv2d_bwd_weight_ws_bytes(ctypes.byref(d))
            ws = torch.empty(max(nb, 16), dtype=torch.uint8, device='cuda')
            t_w = timeit(lambda: check(lib.fcd_conv2d_bwd_weight(ctypes.byref(d), ops._p(x), ops._p(dy), None, ops._p(dw),
                                                                 ops._p(ws), ws.numel(), s)))
            res += ' %8.1f' % (flops / t_w / 1e9)
            ms += ' %6.2f' % t_w
            tot['wgrad'] += t_w
        else:
            res += ' %8s' % '-'
        print('%-26s %5d %10.1f | %s   (%s)%s' % (tag, N, flops / 1e9, res, ms, wino))
        del x, y, dy, dx
    print('sum ms:', {k: round(v, 2) for k, v in tot.items()})


if __name__ == '__main__':
    main()
