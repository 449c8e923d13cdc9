"""Debug helper: fwd / dgrad / wgrad of ops.conv2d vs torch CPU fp64 on the layer shapes of a small Segmentor / VGG."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from fcd_gan_pytorch_amd import _ops as ops
shapes = []
for N in (2, 4, 3):
    for (ci, co, s) in [(4, 64, 32), (64, 64, 32), (64, 128, 16), (128, 128, 16), (128, 256, 8), (256, 256, 8), (256, 512, 4),
                        (512, 512, 4), (512, 512, 2), (2048, 1024, 4), (1024, 512, 4), (1024, 512, 8), (512, 256, 8),
                        (512, 256, 16), (256, 128, 16), (256, 128, 32), (128, 128, 32), (66, 70, 9), (130, 40, 5)]:
        shapes.append((N, ci, s, co))
worst = 0
for (N, C, H, K) in shapes:
    g = torch.Generator().manual_seed(N * 1000 + C + H + K)
    x = torch.randn(N, C, H, H, generator=g); w = torch.randn(K, C, 3, 3, generator=g) * (2.0 / (C * 9)) ** 0.5
    b = torch.randn(K, generator=g); gy = torch.randn(N, K, H, H, generator=g)
    xd, wd = x.double().requires_grad_(True), w.double().requires_grad_(True)
    yr = torch.nn.functional.conv2d(xd, wd, b.double(), 1, 1); yr.backward(gy.double())
    xg, wg, bg = x.cuda().requires_grad_(True), w.cuda().requires_grad_(True), b.cuda().requires_grad_(True)
    y = ops.conv2d(xg, wg, bg, 1, 1); y.backward(gy.cuda())
    e = lambda a, r: ((a.detach().cpu().double() - r).abs().max() / r.abs().max()).item()
    ey, ex, ew = e(y, yr.detach()), e(xg.grad, xd.grad), e(wg.grad, wd.grad)
    # per-sample dx error
    per = [e(xg.grad[n], xd.grad[n]) for n in range(N)]
    flag = '  <<<<' if max(ey, ex, ew) > 2e-5 else ''
    worst = max(worst, ey, ex, ew)
    print('N%d C%4d H%2d K%4d  y %.1e dx %.1e dw %.1e  per-sample dx %s%s' % (N, C, H, K, ey, ex, ew, ' '.join('%.0e' % v for v in per), flag))
print('worst', worst)
