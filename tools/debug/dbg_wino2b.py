import ctypes, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from fcd_gan_pytorch_amd import _ops as ops
from fcd_gan_pytorch_amd._lib import lib, check
N, C, H, W, K = 1, 64, 8, 32, 64
x = torch.zeros(N, C, H, W, device='cuda')
for c in range(C):
    x[0, c] = c + 1
w = torch.zeros(K, C, 3, 3, device='cuda')
for k in range(K):
    w[k, k, 1, 1] = 1.0
d = ops._desc(x.shape, w.shape, 1, 1)
y = torch.full((N, K, H, W), float('nan'), device='cuda')
U = ops.wino2_weight(w, 0)
check(lib.fcd_conv2d_fwd_wino2(ctypes.byref(d), ops._p(x), ops._p(U), None, ops._p(y), 0, None, 0.0, None, None, None, ops._stream()))
torch.cuda.synchronize()
print('y[k] at (3,5):', [round(v, 3) for v in y[0, :, 3, 5].cpu().tolist()])
print('y[k] at (0,0):', [round(v, 3) for v in y[0, :, 0, 0].cpu().tolist()])
Uc = U.view(16, 4, 64, 20).cpu()
# U[q][g][l][xi]: for identity, row=16g+(l&15), red=4q+(l>>4) nonzero iff equal
nz = (Uc[..., :16].abs().sum(-1) > 0)
print('nonzero U entries', int(nz.sum()), 'expected 64')
bad = 0
for q in range(16):
    for g in range(4):
        for l in range(64):
            row, red = 16 * g + (l & 15), 4 * q + (l >> 4)
            if bool(nz[q, g, l]) != (row == red):
                bad += 1
print('U placement mismatches', bad, ' pad nonzero', float(Uc[..., 16:].abs().max()))
print('U center xi values for k=0:', Uc[0, 0, 0, :16].tolist())
