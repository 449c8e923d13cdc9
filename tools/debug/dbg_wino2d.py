import ctypes, os, sys
os.environ['FCD_WINO2_MINC'] = '4'
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ['FCD_LIB'] = os.path.join(ROOT, 'fcd_gan_pytorch_amd', 'libfcdgan_dbg.so')
import torch
sys.path.insert(0, ROOT)
from fcd_gan_pytorch_amd import _ops as ops
from fcd_gan_pytorch_amd._lib import lib, check
N, H, W, K, C = 1, 8, 32, 64, 8
x = torch.zeros(N, C, H, W, device='cuda')
for c in range(C):
    x[0, c] = c + 1
w = torch.randn(K, C, 3, 3, device='cuda')
d = ops._desc(x.shape, w.shape, 1, 1)
y = torch.full((N, K, H, W), float('nan'), device='cuda')
U = ops.wino2_weight(w, 0)
check(lib.fcd_conv2d_fwd_wino2(ctypes.byref(d), ops._p(x), ops._p(U), None, ops._p(y), 0, None, 0.0, None, None, None, ops._stream()))
torch.cuda.synchronize()
flat = y.view(-1).cpu()
slab = flat[:5120]
exp = U.cpu()[5120:10240]
print('slab1 == U chunk 1:', torch.equal(slab, exp), 'max diff', (slab - exp).abs().max().item(), 'nan', torch.isnan(slab).sum().item())
bad = (slab != exp).view(20, 256).sum(1).tolist()
print('mismatches per 1-KiB piece:', bad)
patch = flat[5120:5120 + 4 * 416].view(4, 416)
print('patch buf1 plane sums (expect channel 5..8 * 8*32 interior...):', [round(float(patch[c, :360].view(10, 36)[:, :34].sum()), 1) for c in range(4)])
print('patch plane 0 row 1:', patch[0, 36:36 + 36].tolist())
