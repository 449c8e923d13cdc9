#!/usr/bin/env python3
"""How well is the Discriminator-step gradient of Demo_RSSS (13 bands, 256 x 256, 2 pairs) determined in fp32, and which
way of forming the pooled pair difference is the most accurate?  

For >= 16 density maps that differ by <= 2e-5 (the HIP direct plan's, the HIP Winograd plan's, seeded 1e-5 noise draws,
constant offsets) the D-step gradient  d/dtheta [1 + mean D(unchanged) - mean D(changed)]  is evaluated
  * in fp64 on the CPU oracle (the truth for THAT map),
  * on the fp32 CPU oracle (oneDNN),
  * on the HIP kernels with the three pooling modes of Module.Discriminator_SRGAN_simple.POOL_MODE:
      pooled = round 3 (mean of the batch first, difference of rounded means), diff = reference order on ATen fp32 ops,
      fused = ops.pair_gap_diff (reference order, fp64 accumulator, one kernel).
Prints one row per map and the distribution (median / p90 / max) of each column's relative L2 distance to the truth;
--md writes the table as markdown (profiles/r04_parity_d_probe.md)."""
import argparse
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for q in (ROOT, os.path.join(ROOT, 'tests', 'golden'), os.path.join(ROOT, 'tests')):
    sys.path.insert(0, q)
from seeded import seeded_state, seeded_tiles
from oracle import nets as onets
import fcd_gan_pytorch_amd as p
from fcd_gan_pytorch_amd import _lib
import test_gpu_fullsize_bwd as T

ap = argparse.ArgumentParser()
ap.add_argument('--maps', type=int, default=16)
ap.add_argument('--md', default=None)
args = ap.parse_args()

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


def d_hip(cm, mode):
    p.Module.Discriminator_SRGAN_simple.POOL_MODE = mode
    D = p.Module.Discriminator_SRGAN_simple(C); D.load_state_dict(sdD); D.cuda().train()
    opt = p.optim.RMSprop(D.parameters(), lr=5e-5)
    keep = (1 - cm.detach())
    y_unc = y * (1 - region) + x * region
    c_out, nc_out = D.forward_pairs([(x * keep, y * keep), (x * keep, y_unc * keep)])
    opt.zero_grad()
    (1 + nc_out.mean() - c_out.mean()).backward()
    return {k: prm.grad.detach().cpu().double() for k, prm in D.named_parameters()}


cms = {}
for plan, name in ((0, 'hip-direct'), (4, 'hip-winograd')):
    _lib.lib.fcd_conv_wino_set(plan)
    S = p.Module.Segmentor(C, 1, True); S.load_state_dict(sdS); S.cuda().train()
    with torch.no_grad():
        cms[name] = S(x, y)
_lib.lib.fcd_conv_wino_set(4)
base = cms['hip-direct']
i = 0
while len(cms) < args.maps:
    g = torch.Generator(device='cuda').manual_seed(100 + i)
    if i % 4 == 3:
        cms['const%+.0e' % ((i // 4 + 1) * 1e-5 * (-1) ** (i // 4))] = base + (i // 4 + 1) * 1e-5 * (-1) ** (i // 4)
    else:
        cms['noise1e-5#%d' % i] = base + 1e-5 * torch.randn(base.shape, device='cuda', generator=g)
    i += 1

MODES = ('pooled', 'diff', 'fused')
rows = []
for tag, cm in cms.items():
    t, o = d_truth(cm), d_o32(cm)
    ks = [k for k in t if not T.is_pre_bn_bias(k)]
    cat = lambda d: torch.cat([d[k].reshape(-1).double() for k in ks])
    nt = cat(t).norm()
    row = [tag, ((cat(o) - cat(t)).norm() / nt).item()]
    for mode in MODES:
        row.append(((cat(d_hip(cm, mode)) - cat(t)).norm() / nt).item())
    rows.append(row)
    print('%-16s oracle32 %.2e   HIP pooled %.2e   diff %.2e   fused %.2e' % tuple(row), flush=True)
p.Module.Discriminator_SRGAN_simple.POOL_MODE = 'fused'

cols = ['fp32 CPU oracle'] + ['HIP ' + m for m in MODES]
A = np.array([r[1:] for r in rows])
stats = {c: (np.median(A[:, j]), np.percentile(A[:, j], 90), A[:, j].max(), np.exp(np.log(A[:, j]).mean())) for j, c in enumerate(cols)}
for c in cols:
    print('%-18s median %.2e  p90 %.2e  max %.2e  geo-mean %.2e' % ((c,) + stats[c]))
if args.md:
    L = ['# Discriminator-step gradient vs its fp64 value on %d density maps (tools/parity_probe_d.py)' % len(rows), '',
         'Demo_RSSS D step (Demo_RSSS.py:288-305), 13 bands 256x256, 2 pairs, seeded weights; maps differ by <= 2e-5.  Relative L2 distance of the whole',
         'D gradient (conv biases in front of a BatchNorm excluded) to the fp64 gradient evaluated on the SAME map.', '',
         '| map | ' + ' | '.join(cols) + ' |', '|---|' + '---|' * len(cols)]
    for r in rows:
        L.append('| %s | ' % r[0] + ' | '.join('%.2e' % v for v in r[1:]) + ' |')
    L += ['', '| distribution | ' + ' | '.join(cols) + ' |', '|---|' + '---|' * len(cols)]
    for j, nm in enumerate(('median', '90th percentile', 'max', 'geometric mean')):
        L.append('| %s | ' % nm + ' | '.join('%.2e' % stats[c][j] for c in cols) + ' |')
    os.makedirs(os.path.dirname(os.path.abspath(args.md)), exist_ok=True)
    with open(args.md, 'w') as f:
        f.write('\n'.join(L) + '\n')
