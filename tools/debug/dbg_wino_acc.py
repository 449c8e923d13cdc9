"""Debug helper: full-size forward accuracy of the Winograd settings (run once per FCD_WINO value, compare dumps)."""
import os, sys, warnings
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'tests', 'golden')]
import fcd_gan_pytorch_amd as fcd
from fcd_gan_pytorch_amd.synthetic import synthetic_tiles
tag = os.environ.get('FCD_WINO', '4')
C, H, N = 13, 256, 4
torch.manual_seed(0)
netS = fcd.Module.Segmentor(C, 1, True).cuda().train()
with warnings.catch_warnings():
    warnings.simplefilter('ignore')
    crit = fcd.Loss.CGeneratorLoss(channel=C, perception_layer=1, perception_perBand=True, allow_seeded=True).cuda()
x, y, region = (t.cuda() for t in synthetic_tiles(1234, N, C, H, H))
cmap = netS(x, y)
yf = y + 0.05 * torch.randn(y.shape, device='cuda', generator=torch.Generator(device='cuda').manual_seed(1))
gl, sl, pl = crit(y, yf, cmap)
tot = gl + 0.1 * pl + cmap.abs().mean()
tot.backward()
gflat = torch.cat([p.grad.reshape(-1) for p in netS.parameters()])
torch.save(dict(cmap=cmap.detach().cpu(), pl=pl.detach().cpu(), gl=gl.detach().cpu(), g=gflat.cpu()), '/tmp/acc_%s.pt' % tag)
print(tag, 'perception', pl.item(), 'gen', gl.item())
