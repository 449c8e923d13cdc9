"""Debug helper: Winograd path (FCD_WINO=2/4) vs torch fp64 on wide 3x3 layers, fwd / dgrad / pooled variants."""
import os, sys, time
import torch
import torch.nn.functional as F
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from fcd_gan_pytorch_amd import _ops as ops
print('FCD_WINO =', os.environ.get('FCD_WINO'))
for (N, C, H, W, K) in [(2, 256, 16, 16, 256), (1, 512, 9, 14, 128), (3, 256, 33, 20, 384), (2, 320, 8, 12, 160)]:
    g = torch.Generator().manual_seed(C + H + K)
    x = torch.randn(N, C, H, W, generator=g); w = torch.randn(K, C, 3, 3, generator=g) * (2.0 / (C * 9)) ** 0.5
    b = torch.randn(K, generator=g); gy = torch.randn(N, K, H, W, generator=g)
    xd, wd = x.double().requires_grad_(True), w.double().requires_grad_(True)
    yr = F.conv2d(xd, wd, b.double(), 1, 1); yr.backward(gy.double())
    xg, wg, bg = x.cuda().requires_grad_(True), w.cuda().requires_grad_(True), b.cuda().requires_grad_(True)
    y = ops.conv2d(xg, wg, bg, 1, 1); y.backward(gy.cuda())
    e = lambda a, r: ((a.detach().cpu().double() - r).abs().max() / r.abs().max()).item()
    print('N%d C%d %dx%d K%d  y %.1e dx %.1e dw %.1e' % (N, C, H, W, K, e(y, yr.detach()), e(xg.grad, xd.grad), e(wg.grad, wd.grad)))
    # fused relu + pool
    xg2 = x.cuda().requires_grad_(True)
    wf = w.cuda()
    yp = ops.conv2d_relu_maxpool2(xg2, wf, b.cuda())
    xr2 = x.double().requires_grad_(True)
    ypr = F.max_pool2d(F.relu(F.conv2d(xr2, w.double(), b.double(), 1, 1)), 2)
    gp = torch.randn(ypr.shape, generator=g)
    yp.backward(gp.cuda()); ypr.backward(gp.double())
    d = (xg2.grad.cpu().double() - xr2.grad)
    print('      pool: y %.1e  dx rel-L2 %.1e' % (e(yp, ypr.detach()), (d.norm() / xr2.grad.norm()).item()))
    # relu-mask dgrad
    xg3 = x.cuda().requires_grad_(True)
    y3 = ops.conv2d(xg3, wf, b.cuda(), 1, 1, relu=True); y3.backward(gy.cuda())
    xr3 = x.double().requires_grad_(True)
    F.relu(F.conv2d(xr3, w.double(), b.double(), 1, 1)).backward(gy.double())
    d = (xg3.grad.cpu().double() - xr3.grad)
    print('      relu: dx rel-L2 %.1e' % (d.norm() / xr3.grad.norm()).item())
