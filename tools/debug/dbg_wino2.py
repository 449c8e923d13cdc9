import ctypes, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from fcd_gan_pytorch_amd import _ops as ops
from fcd_gan_pytorch_amd._lib import lib, check
torch.manual_seed(0)
N, C, H, W, K = 1, 64, 8, 32, 64
for name in ('identity', 'random'):
    x = torch.randn(N, C, H, W, device='cuda')
    if name == 'identity':
        w = torch.zeros(K, C, 3, 3, device='cuda')
        for k in range(K):
            w[k, k, 1, 1] = 1.0
    else:
        w = torch.randn(K, C, 3, 3, device='cuda') * 0.1
    d = ops._desc(x.shape, w.shape, 1, 1)
    y = torch.full((N, K, H, W), float('nan'), device='cuda')
    U = ops.wino2_weight(w, 0)
    print(name, 'U finite', torch.isfinite(U).all().item(), U.shape, 'plan', lib.fcd_conv_wino2_plan(ctypes.byref(d), 0))
    check(lib.fcd_conv2d_fwd_wino2(ctypes.byref(d), ops._p(x), ops._p(U), None, ops._p(y), 0, None, 0.0, None, None, None, ops._stream()))
    torch.cuda.synchronize()
    ref = torch.nn.functional.conv2d(x.double().cpu(), w.double().cpu(), padding=1)
    err = (y.cpu().double() - ref).abs()
    print(name, 'nan count', torch.isnan(y).sum().item(), 'max err', err[~torch.isnan(err)].max().item() if (~torch.isnan(err)).any() else None)
    bad = (err > 1e-4) | torch.isnan(err)
    print(' bad by k (first 64):', bad.sum(dim=(0, 2, 3)).tolist())
    print(' bad by row:', bad.sum(dim=(0, 1, 3)).tolist())
    print(' bad by col:', bad.sum(dim=(0, 1, 2)).tolist())
    if name == 'identity':
        print(' y[0,0,:4,:6]', y[0, 0, :4, :6].cpu().tolist())
        print(' x[0,0,:4,:6]', x[0, 0, :4, :6].cpu().tolist())
