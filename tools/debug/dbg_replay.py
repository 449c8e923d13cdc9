"""Debug helper: record every conv op of one criterion test with one library build, replay each op in
isolation with another build and compare (separates real kernel differences from kink-flip propagation).
usage: dbg_replay.py record|replay <tag> <file>"""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'tests', 'golden'), os.path.join(ROOT, 'tests')]
from seeded import seeded_state, seeded_tiles
from oracle import nets as onets
import fcd_gan_pytorch_amd as p
from fcd_gan_pytorch_amd import _ops as ops
mode, tag, path = sys.argv[1:4]
CRIT = {'cnet_pb': ('CNetLoss', 4, 1, True), 'cnet_rgb2': ('CNetLoss', 3, 2, False),
        'cgen_rgb': ('CGeneratorLoss', 3, 1, False)}
if mode == 'record':
    z = np.load(os.path.join(ROOT, 'tests', 'golden', 'losses.npz'))
    seed, cseed, N, C, H, W, allc = [int(v) for v in z[tag + '/meta']]
    cls, channel, layer, pb = CRIT[tag]
    crit = getattr(p.Loss, cls)(channel=channel, perception_layer=layer, perception_perBand=pb)
    crit.loss_perception.net.load_state_dict(seeded_state(onets.vgg_spec(), 4242))
    crit.cuda()
    rec = []
    c2, cp = ops.conv2d, ops.conv2d_relu_maxpool2
    def wrap(fn, kind):
        def f(x, w, b=None, *a, **k):
            e = dict(kind=kind, x=x.detach().cpu(), w=w.detach().cpu(), b=None if b is None else b.detach().cpu(), a=a, k=k)
            xx = x if x.requires_grad else x.detach().requires_grad_(True)
            out = fn(xx, w, b, *a, **k)
            e['out'] = out.detach().cpu()
            out.register_hook(lambda g: e.__setitem__('dout', g.detach().cpu()))
            xx.register_hook(lambda g: e.__setitem__('din', g.detach().cpu()))
            rec.append(e)
            return out
        return f
    ops.conv2d = wrap(c2, 'conv'); ops.conv2d_relu_maxpool2 = wrap(cp, 'pool')
    p.Loss.ops.conv2d = ops.conv2d; p.Loss.ops.conv2d_relu_maxpool2 = ops.conv2d_relu_maxpool2
    t, g, _ = seeded_tiles(seed, N, C, H, W)
    rng = np.random.default_rng([555, cseed])
    cmap = torch.from_numpy(rng.uniform(0.02, 0.98, (N, 1, H, W)).astype(np.float32))
    tg, gg, cg = t.cuda(), g.cuda().requires_grad_(True), cmap.cuda().requires_grad_(True)
    vals = crit(tg, gg, cg)
    sum(w * v for w, v in zip([1.0, 0.3, 0.7, 0.2], vals)).backward()
    torch.save(rec, path)
    print('recorded', len(rec), 'ops')
else:
    rec = torch.load(path)
    for i, e in enumerate(rec):
        fn = ops.conv2d if e['kind'] == 'conv' else ops.conv2d_relu_maxpool2
        x = e['x'].cuda().requires_grad_(True)
        out = fn(x, e['w'].cuda(), None if e['b'] is None else e['b'].cuda(), *e['a'], **e['k'])
        do = (out.detach().cpu() - e['out']).abs()
        line = 'op %2d %-4s x%s w%s args %s %s | out max %.1e (scale %.1e) n>1e-4: %d' % (
            i, e['kind'], tuple(e['x'].shape), tuple(e['w'].shape), e['a'], e['k'], do.max().item(), e['out'].abs().max().item(),
            int((do > 1e-4 * e['out'].abs().max()).sum()))
        if 'dout' in e and 'din' in e:
            out.backward(e['dout'].cuda())
            dd = (x.grad.cpu() - e['din']).abs()
            sc = e['din'].abs().max().item()
            line += ' | din max %.1e (scale %.1e) n>1e-4: %d of %d' % (dd.max().item(), sc, int((dd > 1e-4 * sc).sum()), dd.numel())
        print(line)
