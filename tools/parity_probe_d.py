#!/usr/bin/env python3
"""How well is the Discriminator-step gradient of Demo_RSSS (13 bands, 256 x 256, 2 pairs) determined in fp32?
Evaluates it on the HIP kernels, on the CPU oracle (oneDNN fp32) and in fp64 for four density maps that differ by <= 2e-5
(the HIP direct plan's, the HIP Winograd plan's, + 1e-5 noise, + 1e-5 constant) -- profiles/r03_parity_fullsize.md."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for q in (ROOT, os.path.join(ROOT, 'tests', 'golden'), os.path.join(ROOT, 'tests')):
    sys.path.insert(0, q)
from seeded import seeded_state, seeded_tiles
from oracle import nets as onets
import fcd_gan_pytorch_amd as p
from fcd_gan_pytorch_amd import _lib
import test_gpu_fullsize_bwd as T
C, N, H = 13, 2, 256
sdD = seeded_state(onets.discriminator_spec(C), 13)
sdS = seeded_state(onets.segmentor_spec(C, 1, True), 12)
xc, yc, rc = seeded_tiles(21, N, C, H, H)
x, y, region = xc.cuda(), yc.cuda(), rc.cuda()
def d_truth(cm):
    xd, yd, rd = xc.double(), yc.double(), rc.double()
    keep = 1 - cm.detach().cpu().double()
    return T._d_step_fp64(sdD, (xd * keep, yd * keep), (xd * keep, (yd * (1 - rd) + xd * rd) * keep))
def d_o32(cm):
    oD = onets.clone_state(sdD)
    keep = 1 - cm.detach().cpu()
    c = onets.discriminator(oD, xc * keep, yc * keep, train=True)
    nc = onets.discriminator(oD, xc * keep, (yc * (1 - rc) + xc * rc) * keep, train=True)
    (1 + nc.mean() - c.mean()).backward()
    return {k: oD[k].grad.detach() for k in onets.param_keys(oD)}
cms = {}
for plan in (0, 4):
    _lib.lib.fcd_conv_wino_set(plan)
    S = p.Module.Segmentor(C, 1, True); S.load_state_dict(sdS); S.cuda().train()
    with torch.no_grad():
        cms[plan] = S(x, y)
_lib.lib.fcd_conv_wino_set(0)
g = torch.Generator(device='cuda').manual_seed(1)
cms['noise'] = cms[0] + 1e-5 * torch.randn(cms[0].shape, device='cuda', generator=g)
cms['smooth'] = cms[0] + 1e-5
names = None
for tag, cm in cms.items():
    D = p.Module.Discriminator_SRGAN_simple(C); D.load_state_dict(sdD); D.cuda().train()
    opt = p.optim.RMSprop(D.parameters(), lr=5e-5)
    keep = (1 - cm.detach())
    y_unc = y * (1 - region) + x * region
    c_out, nc_out = D.forward_pairs([(x * keep, y * keep), (x * keep, y_unc * keep)])
    opt.zero_grad()
    (1 + nc_out.mean() - c_out.mean()).backward()
    gh = {k: prm.grad.detach().cpu().double() for k, prm in D.named_parameters()}
    t, o = d_truth(cm), d_o32(cm)
    ks = [k for k in gh if k.startswith('net.') and k.endswith('weight')]
    cat = lambda d: torch.cat([d[k].reshape(-1).double() for k in ks])
    print(tag, 'net weights: HIP vs G64 %.3e   oracle32 vs G64 %.3e   HIP vs oracle32 %.3e' % (
        ((cat(gh) - cat(t)).norm() / cat(t).norm()).item(), ((cat(o) - cat(t)).norm() / cat(t).norm()).item(),
        ((cat(gh) - cat(o)).norm() / cat(t).norm()).item()))
