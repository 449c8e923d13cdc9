"""Debug helper: dump the gradient arriving at every conv output of a Segmentor case (for A/B between library builds)."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'tests', 'golden'), os.path.join(ROOT, 'tests')]
from seeded import seeded_state, seeded_tiles
from oracle import nets as onets
import fcd_gan_pytorch_amd as p
from fcd_gan_pytorch_amd import _ops as ops
tag = sys.argv[1]
z = np.load(os.path.join(ROOT, 'tests', 'golden', 'modules.npz'))
tile_seed, w_seed, ci, N, C, H, W, train = [int(v) for v in z[tag + '/meta']]
x, y, _ = seeded_tiles(tile_seed, N, C, H, W)
bil = tag[2] == 'b'
m = p.Module.Segmentor(C, 1, bil); spec = onets.segmentor_spec(C, 1, bil)
sd = seeded_state(spec, w_seed); m.load_state_dict(sd); m.cuda().train(bool(train))
rec = {}
_conv2d = ops.conv2d
cnt = [0]
def conv_rec(xx, w, b, st=1, pad=0, relu=False):
    i = cnt[0]; cnt[0] += 1
    if xx.requires_grad:
        xx.register_hook(lambda g, i=i: rec.__setitem__('din%02d' % i, g.detach().cpu().clone()))
    out = _conv2d(xx, w, b, st, pad, relu=relu)
    rec['out%02d' % i] = out.detach().cpu().clone()
    out.register_hook(lambda g, i=i: rec.__setitem__('dout%02d' % i, g.detach().cpu().clone()))
    return out
ops.conv2d = conv_rec
xg, yg = x.cuda().requires_grad_(True), y.cuda().requires_grad_(True)
o = m(xg, yg)
rng = np.random.default_rng([991, ci])
pr = torch.from_numpy(rng.standard_normal(tuple(o.shape)).astype(np.float32))
(o * pr.cuda()).sum().backward()
rec['dx'] = xg.grad.cpu(); rec['dy'] = yg.grad.cpu()
for k, v in m.named_parameters():
    rec['dw.' + k] = v.grad.cpu()
torch.save(rec, os.path.join(ROOT, 'gpurun_out', 'rec_%s_%s.pt' % (tag, os.environ.get('TAGLIB', 'new'))))
print('saved', len(rec))
