import ctypes, os, sys
os.environ['FCD_WINO2_MINC'] = '4'
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if len(sys.argv) > 1:
    os.environ['FCD_LIB'] = os.path.join(ROOT, 'fcd_gan_pytorch_amd', sys.argv[1])
import torch
sys.path.insert(0, ROOT)
from fcd_gan_pytorch_amd import _ops as ops
from fcd_gan_pytorch_amd._lib import lib, check
torch.manual_seed(1)
for (N, C, H, W, K) in ((1, 8, 8, 32, 64), (1, 64, 8, 32, 64), (2, 64, 40, 56, 64)):
    x = torch.randn(N, C, H, W, device='cuda')
    w = torch.randn(K, C, 3, 3, device='cuda') * 0.1
    d = ops._desc(x.shape, w.shape, 1, 1)
    y = torch.full((N, K, H, W), float('nan'), device='cuda')
    U = ops.wino2_weight(w, 0)
    check(lib.fcd_conv2d_fwd_wino2(ctypes.byref(d), ops._p(x), ops._p(U), None, ops._p(y), 0, None, 0.0, None, None, None, ops._stream()))
    torch.cuda.synchronize()
    ref = torch.nn.functional.conv2d(x.double().cpu(), w.double().cpu(), padding=1)
    err = (y.cpu().double() - ref).abs()
    print((N, C, H, W, K), 'nan', int(torch.isnan(y).sum()), 'max err', float(err[~torch.isnan(err)].max()), 'ref max', float(ref.abs().max()))
