"""Debug helper: rel-L2 of the input gradients of one tests/golden module case vs the oracle."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'tests', 'golden'), os.path.join(ROOT, 'tests')]
from seeded import seeded_state, seeded_tiles


def probe_like(shape, seed):
    rng = np.random.default_rng([991, seed])
    return torch.from_numpy(rng.standard_normal(tuple(shape)).astype(np.float32))


from oracle import nets as onets
import fcd_gan_pytorch_amd as p
tag = sys.argv[1]
z = np.load(os.path.join(ROOT, 'tests', 'golden', 'modules.npz'))
tile_seed, w_seed, ci, N, C, H, W, train = [int(v) for v in z[tag + '/meta']]
x, y, _ = seeded_tiles(tile_seed, N, C, H, W)
bil = tag[2] == 'b'
m = p.Module.Segmentor(C, 1, bil); spec = onets.segmentor_spec(C, 1, bil)
sd = seeded_state(spec, w_seed); m.load_state_dict(sd); m.cuda().train(bool(train))
xg, yg = x.cuda().requires_grad_(True), y.cuda().requires_grad_(True)
o = m(xg, yg); pr = probe_like(o.shape, ci)
(o * pr.cuda()).sum().backward()
osd = onets.clone_state(sd)
xr, yr = x.clone().requires_grad_(True), y.clone().requires_grad_(True)
orf = onets.segmentor(osd, xr, yr, train=bool(train), bilinear=bil)
(orf * pr).sum().backward()
d = (xg.grad.cpu() - xr.grad).double()
print(tag, 'out Linf', (o.detach().cpu() - orf.detach()).abs().max().item(), 'dx rl2', (d.norm() / xr.grad.double().norm()).item(),
      'max abs', d.abs().max().item(), 'n>1e-5', int((d.abs() > 1e-5).sum()), 'of', d.numel())
bad = (d.abs() > 1e-5).nonzero()
if len(bad):
    print('bad index min', bad.min(0).values.tolist(), 'max', bad.max(0).values.tolist())
torch.save(xg.grad.cpu(), os.path.join(ROOT, 'gpurun_out', 'dx_%s_%s.pt' % (tag, os.environ.get('TAGLIB', 'new'))))

# ---- ReLU-mask disagreement between the HIP path and the oracle (kink flips) ----
import torch.nn.functional as F
from fcd_gan_pytorch_amd import _ops as ops
masks_o, masks_g = [], []
_relu = F.relu
def rec_relu(t, *a, **k):
    masks_o.append(t.detach().clone())
    return _relu(t, *a, **k)
F.relu = rec_relu
onets.F.relu = rec_relu
_bn_act = ops.bn_act
def rec_bn_act(xx, bn, act, **k):
    pre = _bn_act(xx, bn, ops.ACT_NONE, **{kk: v for kk, v in k.items() if kk != 'slope'})
    masks_g.append(pre.detach().cpu().clone())
    return _bn_act(xx, bn, act, **k)
ops.bn_act = rec_bn_act
p.Module.ops.bn_act = rec_bn_act
m(x.cuda(), y.cuda())
with torch.no_grad():
    onets.segmentor(onets.clone_state(sd, requires_grad=False), x, y, train=bool(train), bilinear=bil)
print('layers', len(masks_o), len(masks_g))
# oracle runs the Siamese branches one after the other; the HIP path batches them
import itertools
go = []
i = 0
while i < len(masks_o):
    go.append(masks_o[i]); i += 1
flips = 0
for li, pg in enumerate(masks_g):
    # find oracle tensors with the same trailing shape
    cands = [t for t in masks_o if t.shape[1:] == pg.shape[1:]]
    tot = torch.cat(cands, 0) if sum(t.shape[0] for t in cands) >= pg.shape[0] else None
    small = (pg.abs() < 1e-5).sum().item()
    print('layer', li, tuple(pg.shape), 'n |pre|<1e-5:', small, 'min |pre|: %.2e' % pg.abs().min().item())
