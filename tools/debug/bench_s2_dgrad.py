import ctypes, os, sys, torch
sys.path.insert(0, '.')
from fcd_gan_pytorch_amd import _ops as ops
from fcd_gan_pytorch_amd._lib import lib, check
def timeit(fn, it=10):
    fn(); fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(it): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / it
s = ops._stream()
for (N, C, H, K) in ((32, 13, 256, 64), (32, 64, 128, 128), (32, 128, 64, 256), (32, 256, 32, 512), (16, 64, 128, 128)):
    x = torch.randn(N, C, H, H, device='cuda'); w = torch.randn(K, C, 3, 3, device='cuda') * 0.05
    d = ops._desc(x.shape, w.shape, 2, 1)
    dy = torch.randn(N, K, d.P, d.Q, device='cuda'); dx1 = torch.empty_like(x); dx2 = torch.empty_like(x)
    wp, ws2 = ops.packed_weight(w, 1), ops.s2_weight(w)
    t1 = timeit(lambda: check(lib.fcd_conv2d_bwd_data(ctypes.byref(d), ops._p(dy), None, ops._p(wp), ops._p(dx1), s)))
    t2 = timeit(lambda: check(lib.fcd_conv2d_bwd_data_s2(ctypes.byref(d), ops._p(dy), None, ops._p(ws2), ops._p(dx2), s)))
    print((N, C, H, K), 'dilated %.3f ms  sub-pixel %.3f ms  rel diff %.1e' % (t1, t2, (dx1 - dx2).abs().max().item() / dx1.abs().max().item()))
