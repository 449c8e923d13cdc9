"""Randomised cross-check of the Winograd path against the direct kernels (same library, same process,
`fcd_conv_wino_set`): forward, fused ReLU / ReLU+pool, data gradient (plain / gated / pooled), weight + bias gradient
over random batch / channel / map sizes.  Prints the worst relative deviation per quantity."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from fcd_gan_pytorch_amd import _ops as ops
lib = ops.lib
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
worst = {}
def upd(k, a, b, flips=False):
    a, b = a.double(), b.double()
    if flips:      # ReLU / argmax decisions may differ at rounding level: aggregate measure
        e = ((a - b).norm() / b.norm().clamp_min(1e-30)).item()
    else:
        e = ((a - b).abs().max() / b.abs().max().clamp_min(1e-30)).item()
    worst[k] = max(worst.get(k, 0.0), e)
    return e
ncase = int(sys.argv[2]) if len(sys.argv) > 2 else 60
for it in range(ncase):
    N = int(rng.integers(1, 6)); C = int(rng.integers(2, 17)) * 32; K = int(rng.integers(4, 13)) * 32
    H = int(rng.integers(4, 71)); W = int(rng.integers(4, 71))
    g = torch.Generator(device='cuda').manual_seed(it)
    x = torch.randn(N, C, H, W, device='cuda', generator=g)
    w = torch.randn(K, C, 3, 3, device='cuda', generator=g) * (2.0 / (C * 9)) ** 0.5
    b = torch.randn(K, device='cuda', generator=g) * 0.1
    gy = torch.randn(N, K, H, W, device='cuda', generator=g)
    gp = torch.randn(N, K, H // 2, W // 2, device='cuda', generator=g)
    res = {}
    for m in (0, 4):
        prev = lib.fcd_conv_wino_set(m)
        try:
            xa, wa, ba = x.clone().requires_grad_(True), w.clone().requires_grad_(True), b.clone().requires_grad_(True)
            y = ops.conv2d(xa, wa, ba, 1, 1); y.backward(gy)
            xb = x.clone().requires_grad_(True)
            yr = ops.conv2d(xb, w, b, 1, 1, relu=True); yr.backward(gy)
            out = dict(y=y.detach(), dx=xa.grad, dw=wa.grad, db=ba.grad, yr=yr.detach(), dxr=xb.grad)
            if H >= 2 and W >= 2:
                xc = x.clone().requires_grad_(True)
                yp = ops.conv2d_relu_maxpool2(xc, w, b); yp.backward(gp)
                out.update(yp=yp.detach(), dxp=xc.grad)
            res[m] = out
        finally:
            lib.fcd_conv_wino_set(prev)
    line = 'N%d C%3d K%3d %2dx%2d ' % (N, C, K, H, W)
    for k in res[0]:
        e = upd(k, res[4][k], res[0][k], flips=k in ('dxr', 'dxp'))
        line += ' %s %.1e' % (k, e)
    if it < 5 or it % 10 == 0:
        print(line, flush=True)
print('WORST', {k: '%.1e' % v for k, v in worst.items()})
bad = {k: v for k, v in worst.items() if v > (3e-2 if k in ('dxr', 'dxp') else 1e-4)}
print('FAIL' if bad else 'OK', bad)
