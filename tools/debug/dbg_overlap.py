#!/usr/bin/env python3
"""RSSS adversarial steps with the Discriminator step on the side stream (FCD_STEP_OVERLAP=1) vs in line (default):
per-iteration losses and final weight checksums must be identical."""
import os, sys, copy
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import fcd_gan_pytorch_amd as fcd
from fcd_gan_pytorch_amd import steps, optim
from fcd_gan_pytorch_amd.synthetic import synthetic_tiles

def run(flag, iters=6, n=8, C=13, hw=256):
    os.environ['FCD_STEP_OVERLAP'] = flag
    torch.manual_seed(0)
    dev = torch.device('cuda')
    S = fcd.Module.Segmentor(C, bilinear=True).to(dev).train()
    D = fcd.Module.Discriminator_SRGAN_simple(C).to(dev).train()
    G = fcd.Module.Generator(C).to(dev).eval()
    crit = fcd.Loss.CGeneratorLoss(1, True, allow_seeded=True).to(dev)
    oS, oD = optim.RMSprop(S.parameters(), lr=1e-4), optim.RMSprop(D.parameters(), lr=1e-4)
    out = []
    for it in range(iters):
        x, y, region = (t.to(dev) for t in synthetic_tiles(100 + it, n, C, hw, hw))
        r = steps.rsss_adversarial_step(S, D, G, crit, oS, oD, x, y, region)
        out.append((float(r['d_loss']), float(r['s_loss']), float(r['s_d_loss'])))
    torch.cuda.synchronize()
    cs = (sum(p.double().sum().item() for p in D.parameters()), sum(p.double().sum().item() for p in S.parameters()),
          sum(b.double().sum().item() for b in D.buffers()))
    return out, cs

a = run('0'); b = run('1'); c = run('1')
for i, (u, v, w) in enumerate(zip(a[0], b[0], c[0])):
    print(i, u, v, w)
print(a[1]); print(b[1]); print(c[1])
