#!/usr/bin/env python3
"""Where does the full-size gradient error come from?  For one Demo_RSSS / Demo_WSSS iteration: per-tensor relative L2
of the HIP path's pre-step gradients vs the CPU oracle (direct and Winograd plans, minimal and literal step modes), next
to the ORACLE's own response to a seeded 1e-6 relative perturbation of the weights (the conditioning of the problem)."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests', 'golden')); sys.path.insert(0, os.path.join(ROOT, 'tests'))
from seeded import seeded_state, seeded_tiles          # noqa: E402
from oracle import nets as onets, steps as osteps      # noqa: E402
import fcd_gan_pytorch_amd as p                        # noqa: E402
from fcd_gan_pytorch_amd import _lib                   # noqa: E402

DEV = 'cuda'


def rl2(a, b):
    a, b = a.detach().cpu().double().reshape(-1), b.detach().cpu().double().reshape(-1)
    return ((a - b).norm() / b.norm().clamp_min(1e-30)).item()


def run_oracle(kind, C, N, H, seeds, tiles, eps=0.0):
    sdG, sdS, sdD, sdV = (seeded_state(s, k) for s, k in zip((onets.generator_spec(C), onets.segmentor_spec(C, 1, True),
                                                               onets.discriminator_spec(C), onets.vgg_spec()), seeds))
    n = osteps.Nets(sdG, sdS, sdD, sdV).make_optimizers(kind)
    if eps:
        g = torch.Generator().manual_seed(99)
        with torch.no_grad():
            for sd in (n.S, n.D):
                for k in onets.param_keys(sd):
                    sd[k].mul_(1 + eps * torch.randn(sd[k].shape, generator=g))
    n.capture = {}
    if kind == 'rsss':
        r = osteps.rsss_adversarial_step(n, *tiles)
    else:
        r = osteps.wsss_adversarial_step(n, *tiles)
    return n, r


def run_ours(kind, C, N, H, seeds, tiles, wino, literal):
    _lib.lib.fcd_conv_wino_set(4 if wino else 0)
    sdG, sdS, sdD, sdV = (seeded_state(s, k) for s, k in zip((onets.generator_spec(C), onets.segmentor_spec(C, 1, True),
                                                               onets.discriminator_spec(C), onets.vgg_spec()), seeds))
    netG, netS, netD = p.Module.Generator(C), p.Module.Segmentor(C, 1, True), p.Module.Discriminator_SRGAN_simple(C)
    netG.load_state_dict(sdG); netS.load_state_dict(sdS); netD.load_state_dict(sdD)
    crit = p.Loss.CGeneratorLoss(channel=C, perception_layer=1, perception_perBand=kind == 'rsss', allow_seeded=True)
    crit.loss_perception.net.load_state_dict(sdV)
    for m in (netG, netS, netD, crit):
        m.to(DEV)
    netS.train(); netD.train(); netG.eval()
    lrS, lrD = (5e-5, 5e-5) if kind == 'rsss' else (1e-3, 1e-5)
    oS, oD = p.optim.RMSprop(netS.parameters(), lr=lrS), p.optim.RMSprop(netD.parameters(), lr=lrD)
    st = {}
    oS.pre_step_hooks.append(lambda o: st.__setitem__('S', o.flat_g.detach().cpu().clone()))
    oD.pre_step_hooks.append(lambda o: st.__setitem__('D', o.flat_g.detach().cpu().clone()))
    t = [a.to(DEV) for a in tiles]
    if kind == 'rsss':
        r = p.steps.rsss_adversarial_step(netS, netD, netG, crit, oS, oD, *t, literal=literal)
    else:
        r = p.steps.wsss_adversarial_step(netS, netD, netG, crit, oS, oD, *t, literal=literal)
    out = {}
    for which, net in (('S', netS), ('D', netD)):
        off = 0
        for k, prm in net.named_parameters():
            out[(which, k)] = st[which][off:off + prm.numel()].view(prm.shape)
            off += prm.numel()
    return out, r


def main():
    from test_gpu_modules import is_pre_bn_bias
    for kind, C, N, H, seeds, tseeds in (('rsss', 13, 2, 256, (11, 12, 13, 4242), (21,)), ('wsss', 3, 1, 512, (51, 52, 53, 4242), (54, 55))):
        if kind == 'rsss':
            tiles = seeded_tiles(tseeds[0], N, C, H, H)
        else:
            x, y, _ = seeded_tiles(tseeds[0], N, C, H, H)
            xn, yn, _ = seeded_tiles(tseeds[1], N, C, H, H)
            tiles = (x, y, xn, xn + 0.1 * (yn - xn))
        n0, r0 = run_oracle(kind, C, N, H, seeds, tiles)
        n1, r1 = run_oracle(kind, C, N, H, seeds, tiles, eps=1e-6)
        print('== %s  oracle c/d scalars: d_loss %.6f s_d_loss %.6f' % (kind, float(r0['d_loss']), float(r0['s_d_loss'])))
        for which in ('D', 'S'):
            keys = [k for k in n0.capture[which] if not is_pre_bn_bias(k)]
            cat = lambda d: torch.cat([d[k].reshape(-1) for k in keys])
            print('   oracle sensitivity %s (1e-6 weight perturbation): flat rel-L2 %.2e' % (which, rl2(cat(n1.capture[which]), cat(n0.capture[which]))))
        for wino in (False, True):
            for literal in (False, True):
                got, r = run_ours(kind, C, N, H, seeds, tiles, wino, literal)
                line = '   ours %-8s %-8s d_loss %.6f (ref %.6f)' % ('wino' if wino else 'direct', 'literal' if literal else 'minimal',
                                                                  float(r['d_loss']), float(r0['d_loss']))
                for which in ('D', 'S'):
                    keys = [k for k in n0.capture[which] if not is_pre_bn_bias(k)]
                    flat = rl2(torch.cat([got[(which, k)].reshape(-1) for k in keys]), torch.cat([n0.capture[which][k].reshape(-1) for k in keys]))
                    per = sorted(((rl2(got[(which, k)], n0.capture[which][k]), k) for k in keys if n0.capture[which][k].numel() > 1), reverse=True)[:3]
                    line += '  | %s flat %.2e worst %s' % (which, flat, ', '.join('%s %.1e' % (k, e) for e, k in per))
                print(line)
                sys.stdout.flush()


if __name__ == '__main__':
    main()
