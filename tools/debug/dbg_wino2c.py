import ctypes, os, sys
os.environ['FCD_WINO2_MINC'] = '4'
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from fcd_gan_pytorch_amd import _ops as ops
from fcd_gan_pytorch_amd._lib import lib, check
N, H, W, K = 1, 8, 32, 64
for C in (4, 8, 12, 16):
    x = torch.zeros(N, C, H, W, device='cuda')
    for c in range(C):
        x[0, c] = c + 1
    w = torch.zeros(K, C, 3, 3, device='cuda')
    for k in range(K):
        w[k, k % C, 1, 1] = 1.0          # y[k] = x[k % C] = (k % C) + 1
    d = ops._desc(x.shape, w.shape, 1, 1)
    y = torch.full((N, K, H, W), float('nan'), device='cuda')
    U = ops.wino2_weight(w, 0)
    assert lib.fcd_conv_wino2_plan(ctypes.byref(d), 0)
    check(lib.fcd_conv2d_fwd_wino2(ctypes.byref(d), ops._p(x), ops._p(U), None, ops._p(y), 0, None, 0.0, None, None, None, ops._stream()))
    torch.cuda.synchronize()
    print('C=%d y[k] at (3,5):' % C, [int(v) for v in y[0, :, 3, 5].cpu().tolist()])
