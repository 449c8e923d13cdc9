#!/usr/bin/env python3
"""Capture one train step into a hipGraph and print WHERE a non-capturable call sits (traceback of the first failing op)."""
import os, sys, traceback, warnings
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for q in (ROOT, os.path.join(ROOT, 'tests', 'golden')):
    sys.path.insert(0, q)
import torch
import fcd_gan_pytorch_amd as p
from fcd_gan_pytorch_amd.synthetic import synthetic_tiles
C, N, H = int(os.environ.get('C', 4)), 2, int(os.environ.get('H', 176))
dev = torch.device('cuda')
torch.manual_seed(0)
netD, netS, netG = p.Module.Discriminator_SRGAN_simple(C), p.Module.Segmentor(C, bilinear=True), p.Module.Generator(C)
with warnings.catch_warnings():
    warnings.simplefilter('ignore')
    crit = p.Loss.CGeneratorLoss(channel=C, perception_layer=1, perception_perBand=True, allow_seeded=True)
for m in (netD, netS, netG, crit):
    m.to(dev)
netS.train(); netD.train(); netG.eval()
oS, oD = p.optim.RMSprop(netS.parameters(), lr=5e-5), p.optim.RMSprop(netD.parameters(), lr=5e-5)
x, y, region = (t.to(dev) for t in synthetic_tiles(1, N, C, H, H))
mode = sys.argv[1] if len(sys.argv) > 1 else 'full'
if mode == 'full':
    fn = lambda: p.steps.rsss_adversarial_step(netS, netD, netG, crit, oS, oD, x, y, region)
elif mode == 'fwd':
    def fn():
        with torch.no_grad():
            return netS(x, y)
elif mode == 'loss':
    def fn():
        with torch.no_grad():
            return crit(y, x, torch.rand(N, 1, H, H, device=dev))
if mode == 'usss':
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        crit2 = p.Loss.CNetLoss(channel=C, perception_layer=1, perception_perBand=True, allow_seeded=True).to(dev)
    netG.train()
    oG = p.optim.Adam(netG.parameters(), lr=1e-4, betas=(0.9, 0.99))
    oG.use_device_hyper(); oG.write_hyper()
    fn = lambda: p.steps.usss_g_pretrain_step(netG, crit2, oG, x, y)
for _ in range(2):
    fn()
for o in (oS, oD):
    o.use_device_hyper(); o.write_hyper()
torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
try:
    with torch.cuda.graph(g):
        out = fn()
    print('captured OK')
    g.replay(); torch.cuda.synchronize()
    print('replayed OK')
except Exception:
    traceback.print_exc()
