#!/usr/bin/env python3
"""A/B of the Winograd path per layer shape of the headline workload: per-kernel times (input transform, batched GEMM,
output transform) from the library's per-launch event log.  Usage: [FCD_WINO_XB=1] python tools/bench_wino_gemm.py [filter]
Prints one line per shape; with --check also compares the result with the FCD_WINO_XB=1 library behaviour bit for bit
(run once with FCD_WINO_DUMP=path to write reference outputs, once with FCD_WINO_CMP=path)."""
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fcd_gan_pytorch_amd import _lib, _ops as ops          # noqa: E402
from fcd_gan_pytorch_amd._lib import lib, check             # noqa: E402

NB = int(os.environ.get('NB', '8'))
V = 2 * NB * 13
SHAPES = [  # tag, N, C, HW, K
    ('vgg 64->128 @128', V, 64, 128, 128), ('vgg 128->128 @128', V, 128, 128, 128), ('vgg 128->256 @64', V, 128, 64, 256),
    ('vgg 256->256 @64', V, 256, 64, 256), ('vgg 256->512 @32', V, 256, 32, 512), ('vgg 512->512 @32', V, 512, 32, 512),
    ('vgg 512->512 @16', V, 512, 16, 512),
    ('S enc 128->128 @128', 2 * NB, 128, 128, 128), ('S enc 256->256 @64', 2 * NB, 256, 64, 256),
    ('S enc 512->512 @32', 2 * NB, 512, 32, 512), ('S enc 512->512 @16', 2 * NB, 512, 16, 512),
    ('S dec 2048->1024 @32', NB, 2048, 32, 1024), ('S dec 1024->512 @64', NB, 1024, 64, 512),
    ('S dec 512->256 @128', NB, 512, 128, 256), ('S dec 256->128 @256', NB, 256, 256, 128), ('S dec 128->128 @256', NB, 128, 256, 128),
]


def main():
    only = [a for a in sys.argv[1:] if not a.startswith('--')]
    dump, cmp_ = os.environ.get('FCD_WINO_DUMP'), os.environ.get('FCD_WINO_CMP')
    tot = dict(gemm=0.0, xf=0.0)
    for tag, N, C, HW, K in SHAPES:
        if only and not any(o in tag for o in only):
            continue
        g = torch.Generator(device='cuda').manual_seed(1)
        x = torch.randn(N, C, HW, HW, device='cuda', generator=g)
        if os.environ.get('ZERO_X'):
            x.zero_()
        if os.environ.get('SMALL_X'):
            x.mul_(0).add_(1.0)
        w = torch.randn(K, C, 3, 3, device='cuda', generator=g) * 0.05
        b = torch.zeros(K, device='cuda')
        d = ops._desc(x.shape, w.shape, 1, 1)
        m = lib.fcd_conv_wino_plan(ctypes.byref(d), 0)
        if not m:
            continue
        y = torch.empty(N, K, HW, HW, device='cuda')
        U = ops.wino_weight(w, 0, m)
        ws = torch.empty(lib.fcd_conv_wino_ws_bytes(ctypes.byref(d), 0), dtype=torch.uint8, device='cuda')
        s = ops._stream()

        def run():
            check(lib.fcd_conv2d_fwd_wino(ctypes.byref(d), ops._p(x), ops._p(U), ops._p(b), ops._p(y), 0, None, None,
                                          ops._p(ws), ws.numel(), s))
        run(); run()
        torch.cuda.synchronize()
        _lib.prof_read(reset=True)
        lib.fcd_prof_enable(2)
        for _ in range(5):
            run()
        torch.cuda.synchronize()
        lib.fcd_prof_enable(0)
        _lib.prof_read(reset=True)
        det = _lib.prof_detail(reset=True)
        gm = [e for e in det if e['family'] in ('wino_gemm', 'wino_gemm_bf16x6')]
        xf = [e for e in det if e['family'] == 'wino_transform']
        tg = sum(e['ms'] for e in gm) / 5
        tx = sum(e['ms'] for e in xf) / 5
        fl = gm[0]['flops']
        tot['gemm'] += tg; tot['xf'] += tx
        extra = ''
        if dump:
            torch.save(y.cpu(), os.path.join(dump, tag.replace(' ', '_').replace('>', '') + '.pt'))
        if cmp_:
            ref = torch.load(os.path.join(cmp_, tag.replace(' ', '_').replace('>', '') + '.pt'))
            extra = '  bit-equal=%s' % bool(torch.equal(ref, y.cpu()))
        ti = sum(e['ms'] for e in xf if e['tag'].startswith('in ')) / 5
        to = sum(e['ms'] for e in xf if e['tag'].startswith('out ')) / 5
        bi = sum(e['bytes'] for e in xf if e['tag'].startswith('in ')) / 5
        bo = sum(e['bytes'] for e in xf if e['tag'].startswith('out ')) / 5
        print('%-22s gemm %7.3f ms  %6.1f TF (%.2f)   in %7.3f ms %5.0f GB/s   out %7.3f ms %5.0f GB/s%s'
              % (tag, tg, fl / tg / 1e9, fl / tg / 1e9 / 157.3, ti, bi / ti / 1e6, to, bo / to / 1e6, extra))
        del x, y, ws
    print('sum: gemm %.2f ms, transforms %.2f ms' % (tot['gemm'], tot['xf']))


if __name__ == '__main__':
    main()
