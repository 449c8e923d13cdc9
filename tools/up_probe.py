"""Timing of the x2 bilinear upsampling kernels on the decoder's maps (HIP events, median of 20) + a checksum of the results."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from fcd_gan_pytorch_amd import _ops as ops
from fcd_gan_pytorch_amd._lib import lib, check

torch.manual_seed(0)
for NC, H, W in ((8 * 128, 128, 128), (8 * 256, 64, 64), (8 * 512, 32, 32), (8 * 1024, 16, 16), (7, 27, 54)):
    x = torch.randn(NC, H, W, device='cuda')
    y = torch.empty(NC, 2 * H, 2 * W, device='cuda')
    dx = torch.empty_like(x)
    res = []
    for name, fn in (('fwd', lambda: check(lib.fcd_upsample2x_fwd(ops._p(x), ops._p(y), NC, H, W, ops._stream()))),
                     ('bwd', lambda: check(lib.fcd_upsample2x_bwd(ops._p(y), ops._p(dx), NC, H, W, ops._stream())))):
        ts = []
        for r in range(23):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record(); fn(); b.record(); torch.cuda.synchronize()
            if r >= 3:
                ts.append(a.elapsed_time(b) * 1e3)
        ts.sort()
        res.append('%s %.1f us (%.2f TB/s)' % (name, ts[10], 5 * x.numel() * 4 / ts[10] / 1e6))
    import hashlib
    hy = hashlib.sha1(y.cpu().numpy().tobytes()).hexdigest()[:10]
    hd = hashlib.sha1(dx.cpu().numpy().tobytes()).hexdigest()[:10]
    print((NC, H, W), '  '.join(res), 'sha1 y %s dx %s' % (hy, hd), flush=True)
