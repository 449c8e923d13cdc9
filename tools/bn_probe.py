"""Timing probe of the BatchNorm kernels on the Segmentor's largest layer shapes (HIP events, 20 repetitions)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from fcd_gan_pytorch_amd import _ops as ops

def run(shape, groups, reps=20):
    N, C, H, W = shape
    bn = torch.nn.BatchNorm2d(C).cuda().train()
    x = torch.randn(shape, device='cuda', requires_grad=True)
    g = torch.randn(shape, device='cuda')
    out = {}
    for name in ('fwd', 'bwd'):
        ts = []
        for r in range(reps + 3):
            y = ops.bn_act(x, bn, ops.ACT_RELU, groups=groups)
            torch.cuda.synchronize()
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            if name == 'fwd':
                a.record(); y2 = ops.bn_act(x, bn, ops.ACT_RELU, groups=groups); b.record()
            else:
                a.record(); y.backward(g); b.record()
            torch.cuda.synchronize()
            if r >= 3:
                ts.append(a.elapsed_time(b))
        ts.sort()
        out[name] = ts[len(ts) // 2] * 1e3
    mb = N * C * H * W * 4 / 1e6
    print('%s groups=%d  %.0f MB  fwd %.1f us (%.2f TB/s on 3 passes)  bwd %.1f us (%.2f TB/s on 5 passes)  FCD_BN_ROT=%s' % (
        shape, groups, mb, out['fwd'], 3 * mb / out['fwd'], out['bwd'], 5 * mb / out['bwd'], os.environ.get('FCD_BN_ROT')), flush=True)

for shape, gr in (((16, 64, 256, 256), 2), ((8, 128, 256, 256), 1), ((16, 128, 128, 128), 2), ((16, 256, 64, 64), 2), ((8, 256, 128, 128), 1)):
    run(shape, gr)
