#!/usr/bin/env python3
"""Generate the golden fixtures (``*.npz``) by running the REFERENCE ITSELF.

Runs only in the build container (needs /root/reference); the GPU box and the
test-suite only ever see the ``.npz`` outputs.  The reference's Module.py /
Loss.py import torchvision, osgeo and (through CommonFunc) tqdm; torchvision and
GDAL are absent here, so empty ``sys.modules`` stubs are installed first, with a
``vgg16`` stand-in that builds torchvision's cfg-D ``features`` stack (weights are
then overwritten with seeded values: the ImageNet checkpoint is unobtainable
offline => perception parity is pinned for seeded VGG weights only).

Usage:  python tests/golden/gen_golden.py [--only modules|losses|steps|steps2|ckpt]
"""
import argparse
import os
import sys
import types

import numpy as np
import torch
import torch.nn as nn

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)

from seeded import seeded_state, seeded_tiles, summary  # noqa: E402
from oracle import nets as onets  # noqa: E402  (only for the key/shape specs it pins)

REF = '/root/reference'


def install_stubs():
    def mod(name):
        m = types.ModuleType(name)
        sys.modules[name] = m
        return m
    tv = mod('torchvision')
    tv.transforms = mod('torchvision.transforms')
    tv.models = mod('torchvision.models')
    vggm = mod('torchvision.models.vgg')
    tv.models.vgg = vggm

    def vgg16(pretrained=False, **kw):
        layers, cin = [], 3
        for v in onets.VGG_CFG:
            if v == 'M':
                layers.append(nn.MaxPool2d(kernel_size=2, stride=2))
            else:
                layers += [nn.Conv2d(cin, v, kernel_size=3, padding=1), nn.ReLU(inplace=True)]
                cin = v
        holder = nn.Module()
        holder.features = nn.Sequential(*layers)
        return holder
    vggm.vgg16 = vgg16
    og = mod('osgeo')
    for sub in ('gdal', 'ogr', 'osr'):
        setattr(og, sub, mod('osgeo.' + sub))
    mod('cv2')


def load_reference():
    install_stubs()
    sys.path.insert(0, REF)
    import Module as RM  # noqa
    import Loss as RL    # noqa
    import ssim as RS    # noqa
    return RM, RL, RS


def pin_spec(module, spec, what):
    got = {k: tuple(v.shape) for k, v in module.state_dict().items()}
    assert list(got.keys()) == list(spec.keys()), '%s: key order/name mismatch' % what
    assert got == {k: tuple(v) for k, v in spec.items()}, '%s: shape mismatch' % what


def grads_of(module):
    return {k: p.grad for k, p in module.named_parameters()}


def pack_grads(out, tag, named):
    for k, g in named.items():
        out['%s/grad/%s' % (tag, k)] = summary(g)


def pack_buffers(out, tag, module):
    for k, v in module.state_dict().items():
        if 'running_' in k or 'num_batches' in k:
            out['%s/buf/%s' % (tag, k)] = v.detach().double().numpy()


def probe_like(t, seed):
    rng = np.random.default_rng([991, seed])
    return torch.from_numpy(rng.standard_normal(tuple(t.shape)).astype(np.float32))


# ---------------------------------------------------------------- modules
def gen_modules(RM):
    out = {}
    cases = [
        ('G4_32', 'G', dict(C=4), (2, 4, 32, 32), True),
        ('G4_32_eval', 'G', dict(C=4), (2, 4, 32, 32), False),
        ('G13_24x40', 'G', dict(C=13), (1, 13, 24, 40), True),
        ('S4b_32', 'S', dict(C=4, bilinear=True), (2, 4, 32, 32), True),
        ('S4b_40x56', 'S', dict(C=4, bilinear=True), (2, 4, 40, 56), True),
        ('S4b_32_eval', 'S', dict(C=4, bilinear=True), (2, 4, 32, 32), False),
        ('S4t_32', 'S', dict(C=4, bilinear=False), (2, 4, 32, 32), True),
        ('S3t_40x56', 'S', dict(C=3, bilinear=False), (1, 3, 40, 56), True),
        ('D4_32', 'D', dict(C=4), (2, 4, 32, 32), True),
        ('D4_48x40', 'D', dict(C=4), (3, 4, 48, 40), True),
        ('D3_38x50', 'D', dict(C=3), (2, 3, 38, 50), True),
    ]
    for ci, (tag, kind, kw, shape, train) in enumerate(cases):
        N, C, H, W = shape
        x, y, _ = seeded_tiles(100 + ci, N, C, H, W)
        x.requires_grad_(True); y.requires_grad_(True)
        if kind == 'G':
            m = RM.Generator(C); spec = onets.generator_spec(C)
        elif kind == 'S':
            m = RM.Segmentor(C, 1, kw['bilinear']); spec = onets.segmentor_spec(C, 1, kw['bilinear'])
        else:
            m = RM.Discriminator_SRGAN_simple(C); spec = onets.discriminator_spec(C)
        pin_spec(m, spec, tag)
        m.load_state_dict(seeded_state(spec, 1000 + ci))
        m.train(train)
        o = m(x) if kind == 'G' else m(x, y)
        pr = probe_like(o, ci)
        (o * pr).sum().backward()
        out[tag + '/out'] = o.detach().numpy()
        out[tag + '/dx'] = summary(x.grad)
        if kind != 'G':
            out[tag + '/dy'] = summary(y.grad)
        pack_grads(out, tag, grads_of(m))
        if train:
            pack_buffers(out, tag, m)
        out[tag + '/meta'] = np.array([100 + ci, 1000 + ci, ci, N, C, H, W, int(train)], np.int64)
        print('modules', tag, tuple(o.shape), float(o.abs().mean()))
    np.savez_compressed(os.path.join(HERE, 'modules.npz'), **out)


# ----------------------------------------------------------------- losses
def make_cmap(seed, N, H, W, all_changed=None):
    rng = np.random.default_rng([555, seed])
    c = torch.from_numpy(rng.uniform(0.02, 0.98, (N, 1, H, W)).astype(np.float32))
    if all_changed is not None:
        c[all_changed] = 1.0
    return c


def gen_losses(RL, RS):
    out = {}
    vgg_sd = seeded_state(onets.vgg_spec(), 4242)
    # MS-SSIM alone (value + input gradients), even and odd pooling chains
    for tag, (N, C, H, W) in (('msssim_176', (2, 4, 176, 176)), ('msssim_200x184', (1, 3, 200, 184))):
        x, y, _ = seeded_tiles(300 + H, N, C, H, W)
        x.requires_grad_(True); y.requires_grad_(True)
        v = RS.MS_SSIM(data_range=1.0, channel=C)(x, y)
        v.backward()
        out[tag + '/val'] = v.detach().double().numpy()
        out[tag + '/dx'] = summary(x.grad); out[tag + '/dy'] = summary(y.grad)
        out[tag + '/dx_full'] = x.grad[0, 0, ::8, ::8].numpy()
        v2 = RS.SSIM(data_range=1.0, channel=C)(x.detach(), y.detach())
        out[tag + '/ssim_val'] = v2.double().numpy()
        out[tag + '/meta'] = np.array([300 + H, N, C, H, W], np.int64)
        print('losses', tag, float(v))
    # criteria
    for tag, cls, kw, (N, C, H, W), allc in (
            ('cnet_pb', 'CNetLoss', dict(channel=4, perception_layer=1, perception_perBand=True), (2, 4, 176, 176), None),
            ('cnet_rgb2', 'CNetLoss', dict(channel=3, perception_layer=2, perception_perBand=False), (1, 3, 176, 176), None),
            ('cgen_rgb', 'CGeneratorLoss', dict(channel=3, perception_layer=1, perception_perBand=False), (2, 3, 176, 176), None),
            ('cgen_pb_allchanged', 'CGeneratorLoss', dict(channel=4, perception_layer=1, perception_perBand=True), (2, 4, 176, 176), 1)):
        crit = getattr(RL, cls)(**kw)
        crit.loss_perception.net.load_state_dict(vgg_sd)
        t, g, _ = seeded_tiles(400 + len(tag), N, C, H, W)
        cmap = make_cmap(len(tag), N, H, W, allc)
        g.requires_grad_(True); cmap.requires_grad_(True)
        vals = crit(t, g, cmap)
        wts = [1.0, 0.3, 0.7, 0.2][:len(vals)]
        tot = sum(w * v for w, v in zip(wts, vals))
        tot.backward()
        out[tag + '/vals'] = np.array([float(v) for v in vals], np.float64)
        out[tag + '/dgen'] = summary(g.grad); out[tag + '/dcmap'] = summary(cmap.grad)
        out[tag + '/meta'] = np.array([400 + len(tag), len(tag), N, C, H, W, -1 if allc is None else allc], np.int64)
        print('losses', tag, out[tag + '/vals'])
    # region loss with one empty region
    cm = make_cmap(77, 3, 40, 48); cm.requires_grad_(True)
    reg = torch.zeros(3, 1, 40, 48); reg[0, :, 5:20, 8:30] = 1; reg[2, :, 0:40, 0:10] = 1
    a = RL.region_loss(cm, reg, nn.L1Loss()); b = RL.region_loss(cm, 1 - reg, nn.MSELoss())
    (a + 2 * b).backward()
    out['region/vals'] = np.array([float(a), float(b)], np.float64)
    out['region/dcmap'] = summary(cm.grad)
    np.savez_compressed(os.path.join(HERE, 'losses.npz'), **out)


# ------------------------------------------------------------------ steps
def weights_summary(out, tag, module):
    for k, v in module.state_dict().items():
        out['%s/%s' % (tag, k)] = summary(v.float()) if v.is_floating_point() else v.double().numpy()


def build_nets(RM, RL, C, crit_cls, crit_kw, seedbase, bilinear=True):
    G = RM.Generator(C); G.load_state_dict(seeded_state(onets.generator_spec(C), seedbase + 1))
    S = RM.Segmentor(C, 1, bilinear); S.load_state_dict(seeded_state(onets.segmentor_spec(C, 1, bilinear), seedbase + 2))
    D = RM.Discriminator_SRGAN_simple(C); D.load_state_dict(seeded_state(onets.discriminator_spec(C), seedbase + 3))
    crit = getattr(RL, crit_cls)(**crit_kw)
    crit.loss_perception.net.load_state_dict(seeded_state(onets.vgg_spec(), 4242))
    return G, S, D, crit


def gen_steps(RM, RL):
    out = {}
    H = W = 176
    # --- RSSS adversarial iteration, literal transcription of Demo_RSSS.py:285-332
    C, N = 4, 2
    G, S, D, crit = build_nets(RM, RL, C, 'CGeneratorLoss', dict(channel=C, perception_layer=1, perception_perBand=True), 7000)
    S.train(); D.train(); G.eval()
    oS = torch.optim.RMSprop(S.parameters(), lr=5e-5); oD = torch.optim.RMSprop(D.parameters(), lr=5e-5)
    x, y, region = seeded_tiles(7100, N, C, H, W)
    for it in range(2):
        cmap = S(x, y); cmask = cmap
        x_mask = x * (1 - cmask.repeat((1, C, 1, 1))); y_mask = y * (1 - cmask.repeat((1, C, 1, 1)))
        c_out = D(x_mask, y_mask)
        x_unc = x; y_unc = y * (1 - region) + x * region
        x_unc = x_unc * (1 - cmask.repeat((1, C, 1, 1))); y_unc = y_unc * (1 - cmask.repeat((1, C, 1, 1)))
        nc_out = D(x_unc, y_unc)
        oD.zero_grad(); d_loss = 1 + nc_out.mean() - c_out.mean(); d_loss.backward(retain_graph=True); oD.step()
        c_out = D(x_mask, y_mask)
        y_fake = G(x)
        generator_loss, ssim_loss, perception_loss = crit(y, y_fake, cmap)
        g_loss = generator_loss + 0.1 * perception_loss + 0 * ssim_loss
        l1_loss = RL.region_loss(cmap, region, nn.L1Loss()); s_d_loss = c_out.mean()
        r_loss = RL.region_loss(cmap, 1 - region, nn.MSELoss())
        s_loss = 1 * s_d_loss + 0.02 * l1_loss + 0.5 * g_loss + 2 * r_loss
        oS.zero_grad(); s_loss.backward(); oS.step()
        out['rsss/it%d/scalars' % it] = np.array([float(v) for v in (d_loss, s_loss, s_d_loss, g_loss, l1_loss, r_loss, generator_loss, ssim_loss, perception_loss)], np.float64)
        out['rsss/it%d/cmap' % it] = cmap.detach()[:, :, ::4, ::4].numpy()
        out['rsss/it%d/cmap_sum' % it] = summary(cmap)
        print('steps rsss it', it, out['rsss/it%d/scalars' % it])
    weights_summary(out, 'rsss/S', S); weights_summary(out, 'rsss/D', D); weights_summary(out, 'rsss/G', G)
    out['rsss/meta'] = np.array([7000, 7100, N, C, H, W], np.int64)

    # --- USSS joint iteration, Demo_USSS.py:310-341
    C, N = 4, 1
    G, S, D, crit = build_nets(RM, RL, C, 'CNetLoss', dict(channel=C, perception_layer=1, perception_perBand=True), 8000)
    S.train(); G.train()
    oG = torch.optim.Adam(G.parameters(), lr=2e-4, betas=(0.9, 0.99)); oS = torch.optim.Adam(S.parameters(), lr=2e-4, betas=(0.9, 0.99))
    x, y, _ = seeded_tiles(8100, N, C, H, W)
    for it in range(2):
        oG.zero_grad()
        y_fake = G(x); cmap = S(x, y)
        generator_loss, l1_loss, perception_loss, ssim_loss = crit(y, y_fake, cmap)
        Loss = generator_loss + 0.4 * perception_loss + 0 * ssim_loss
        Loss.backward(retain_graph=True)
        NetLoss = generator_loss + 0.65 * l1_loss + 0.4 * perception_loss + 0 * ssim_loss
        oS.zero_grad(); NetLoss.backward()
        oG.step(); oS.step()
        out['usss/it%d/scalars' % it] = np.array([float(v) for v in (Loss, NetLoss, generator_loss, l1_loss, perception_loss, ssim_loss)], np.float64)
        out['usss/it%d/cmap_sum' % it] = summary(cmap)
        print('steps usss it', it, out['usss/it%d/scalars' % it])
    weights_summary(out, 'usss/S', S); weights_summary(out, 'usss/G', G)
    out['usss/meta'] = np.array([8000, 8100, N, C, H, W], np.int64)

    # --- WSSS adversarial iteration, Demo_WSSS.py:249-323
    C, N = 3, 1
    G, S, D, crit = build_nets(RM, RL, C, 'CGeneratorLoss', dict(channel=C, perception_layer=1, perception_perBand=False), 9000)
    S.train(); D.train(); G.eval()
    oS = torch.optim.RMSprop(S.parameters(), lr=1e-3); oD = torch.optim.RMSprop(D.parameters(), lr=1e-5)
    x, y, _ = seeded_tiles(9100, N, C, H, W)
    x_nc, _, _ = seeded_tiles(9200, N, C, H, W)
    y_nc = x_nc + 0.05 * seeded_tiles(9300, N, C, H, W)[0]
    for it in range(1):
        cmap = S(x, y); cmask = cmap
        x_mask = x * (1 - cmask.repeat((1, C, 1, 1))); y_mask = y * (1 - cmask.repeat((1, C, 1, 1)))
        c_out = D(x_mask, y_mask)
        ncmap = S(x_nc, y_nc)
        x_mask_nc = x_nc * (1 - cmask.repeat((1, C, 1, 1))); y_mask_nc = y_nc * (1 - cmask.repeat((1, C, 1, 1)))
        nc_out = D(x_mask_nc, y_mask_nc)
        oD.zero_grad(); d_loss = 1 + nc_out.mean() - c_out.mean(); d_loss.backward(retain_graph=True); oD.step()
        nc_loss = torch.mean(torch.pow(ncmap, 2))
        c_out = D(x_mask, y_mask)
        y_fake = G(x)
        generator_loss, ssim_loss, perception_loss = crit(y, y_fake, cmap)
        g_loss = generator_loss + 0.5 * perception_loss + 0 * ssim_loss
        l1_loss = torch.mean(abs(cmap)); s_d_loss = c_out.mean()
        s_loss = 1 * s_d_loss + 1.6 * l1_loss + 0.2 * g_loss + 1.5 * nc_loss
        oS.zero_grad(); s_loss.backward(); oS.step()
        out['wsss/it%d/scalars' % it] = np.array([float(v) for v in (d_loss, s_loss, s_d_loss, g_loss, l1_loss, nc_loss, generator_loss, ssim_loss, perception_loss)], np.float64)
        out['wsss/it%d/cmap_sum' % it] = summary(cmap)
        print('steps wsss it', it, out['wsss/it%d/scalars' % it])
    weights_summary(out, 'wsss/S', S); weights_summary(out, 'wsss/D', D)
    out['wsss/meta'] = np.array([9000, 9100, N, C, H, W], np.int64)
    np.savez_compressed(os.path.join(HERE, 'steps.npz'), **out)


def grad_summaries(out, tag, module):
    for k, p in module.named_parameters():
        out['%s/%s' % (tag, k)] = summary(p.grad)


def ref_adjust_lr():
    """The reference's own adjust_learning_rate (CommonFunc.py:23-37); CommonFunc imports osgeo / cv2 at module
    level, which the stubs satisfy."""
    import CommonFunc as RC  # noqa
    return RC.adjust_learning_rate


def gen_steps_extra(RM, RL):
    """steps2.npz: (1) the gradients each optimizer steps on in the FIRST iteration of the three demos (before any
    sign-like RMSprop / Adam update can amplify rounding noise) -- pins the oracle's backward bit-for-bit;
    (2) a 6-iteration Demo_RSSS trajectory with the reference's adjust_learning_rate in the loop
    (Demo_RSSS.py:248-249: one "epoch" per iteration here).  The trajectory runs at epochs 40..45 of the schedule
    (decay branch, lr_S ~1.4e-6): at the warm-up rates (1e-4..1e-3 with RMSprop's sign-like first steps) the
    reference's own trajectory is chaotic -- the CPU oracle, same ATen ops with a 5e-6 relative difference in
    gradient summation order, is 0.1 off in the density map after 3 iterations -- and pins nothing."""
    out = {}
    H = W = 176
    adjust = ref_adjust_lr()
    # ---- RSSS: 6 iterations, LR schedule per iteration, gradients of iteration 0
    C, N = 4, 2
    G, S, D, crit = build_nets(RM, RL, C, 'CGeneratorLoss', dict(channel=C, perception_layer=1, perception_perBand=True), 7000)
    S.train(); D.train(); G.eval()
    oS = torch.optim.RMSprop(S.parameters(), lr=5e-5); oD = torch.optim.RMSprop(D.parameters(), lr=5e-5)
    x, y, region = seeded_tiles(7100, N, C, H, W)
    lrs = []
    EP0 = 40
    for it in range(6):
        adjust(oS, EP0 + it, lr_start=1e-4, lr_max=1e-3, lr_warm_up_epoch=5)                      # Demo_RSSS.py:248
        adjust(oD, EP0 + it, lr_start=5e-6, lr_max=5e-5, lr_min=5e-7, lr_warm_up_epoch=5)         # Demo_RSSS.py:249
        lrs.append([oS.param_groups[0]['lr'], oD.param_groups[0]['lr']])
        cmap = S(x, y); cmask = cmap
        x_mask = x * (1 - cmask.repeat((1, C, 1, 1))); y_mask = y * (1 - cmask.repeat((1, C, 1, 1)))
        c_out = D(x_mask, y_mask)
        x_unc = x; y_unc = y * (1 - region) + x * region
        x_unc = x_unc * (1 - cmask.repeat((1, C, 1, 1))); y_unc = y_unc * (1 - cmask.repeat((1, C, 1, 1)))
        nc_out = D(x_unc, y_unc)
        oD.zero_grad(); d_loss = 1 + nc_out.mean() - c_out.mean(); d_loss.backward(retain_graph=True)
        if it == 0:
            grad_summaries(out, 'rsss/it0/gradD', D)
        oD.step()
        c_out = D(x_mask, y_mask)
        y_fake = G(x)
        generator_loss, ssim_loss, perception_loss = crit(y, y_fake, cmap)
        g_loss = generator_loss + 0.1 * perception_loss + 0 * ssim_loss
        l1_loss = RL.region_loss(cmap, region, nn.L1Loss()); s_d_loss = c_out.mean()
        r_loss = RL.region_loss(cmap, 1 - region, nn.MSELoss())
        s_loss = 1 * s_d_loss + 0.02 * l1_loss + 0.5 * g_loss + 2 * r_loss
        oS.zero_grad(); s_loss.backward()
        if it == 0:
            grad_summaries(out, 'rsss/it0/gradS', S)
        oS.step()
        out['traj/it%d/scalars' % it] = np.array([float(v) for v in (d_loss, s_loss, s_d_loss, g_loss, l1_loss, r_loss, generator_loss, ssim_loss, perception_loss)], np.float64)
        out['traj/it%d/cmap' % it] = cmap.detach()[:, :, ::4, ::4].numpy()
        print('traj rsss it', it, out['traj/it%d/scalars' % it])
    out['traj/lrs'] = np.array(lrs, np.float64)
    weights_summary(out, 'traj/S', S); weights_summary(out, 'traj/D', D)
    out['traj/meta'] = np.array([7000, 7100, N, C, H, W, 6, EP0], np.int64)

    # ---- USSS joint, iteration 0 gradients (G: grad(Loss)+grad(NetLoss); S: grad(NetLoss))
    C, N = 4, 1
    G, S, D, crit = build_nets(RM, RL, C, 'CNetLoss', dict(channel=C, perception_layer=1, perception_perBand=True), 8000)
    S.train(); G.train()
    oG = torch.optim.Adam(G.parameters(), lr=2e-4, betas=(0.9, 0.99)); oS = torch.optim.Adam(S.parameters(), lr=2e-4, betas=(0.9, 0.99))
    x, y, _ = seeded_tiles(8100, N, C, H, W)
    oG.zero_grad()
    y_fake = G(x); cmap = S(x, y)
    generator_loss, l1_loss, perception_loss, ssim_loss = crit(y, y_fake, cmap)
    Loss = generator_loss + 0.4 * perception_loss + 0 * ssim_loss
    Loss.backward(retain_graph=True)
    NetLoss = generator_loss + 0.65 * l1_loss + 0.4 * perception_loss + 0 * ssim_loss
    oS.zero_grad(); NetLoss.backward()
    grad_summaries(out, 'usss/it0/gradG', G); grad_summaries(out, 'usss/it0/gradS', S)

    # ---- WSSS, iteration 0 gradients
    C, N = 3, 1
    G, S, D, crit = build_nets(RM, RL, C, 'CGeneratorLoss', dict(channel=C, perception_layer=1, perception_perBand=False), 9000)
    S.train(); D.train(); G.eval()
    oS = torch.optim.RMSprop(S.parameters(), lr=1e-3); oD = torch.optim.RMSprop(D.parameters(), lr=1e-5)
    x, y, _ = seeded_tiles(9100, N, C, H, W)
    x_nc, _, _ = seeded_tiles(9200, N, C, H, W)
    y_nc = x_nc + 0.05 * seeded_tiles(9300, N, C, H, W)[0]
    cmap = S(x, y); cmask = cmap
    x_mask = x * (1 - cmask.repeat((1, C, 1, 1))); y_mask = y * (1 - cmask.repeat((1, C, 1, 1)))
    c_out = D(x_mask, y_mask)
    ncmap = S(x_nc, y_nc)
    x_mask_nc = x_nc * (1 - cmask.repeat((1, C, 1, 1))); y_mask_nc = y_nc * (1 - cmask.repeat((1, C, 1, 1)))
    nc_out = D(x_mask_nc, y_mask_nc)
    oD.zero_grad(); d_loss = 1 + nc_out.mean() - c_out.mean(); d_loss.backward(retain_graph=True)
    grad_summaries(out, 'wsss/it0/gradD', D)
    oD.step()
    nc_loss = torch.mean(torch.pow(ncmap, 2))
    c_out = D(x_mask, y_mask)
    y_fake = G(x)
    generator_loss, ssim_loss, perception_loss = crit(y, y_fake, cmap)
    g_loss = generator_loss + 0.5 * perception_loss + 0 * ssim_loss
    l1_loss = torch.mean(abs(cmap)); s_d_loss = c_out.mean()
    s_loss = 1 * s_d_loss + 1.6 * l1_loss + 0.2 * g_loss + 1.5 * nc_loss
    oS.zero_grad(); s_loss.backward()
    grad_summaries(out, 'wsss/it0/gradS', S)
    np.savez_compressed(os.path.join(HERE, 'steps2.npz'), **out)


def gen_checkpoint(RM):
    """Checkpoint interchange (Demo_RSSS.py:167-171,507-514; Demo_USSS.py:477-481): ``netG_ref.pkl`` is
    ``torch.save(netG.state_dict())`` of the REFERENCE Generator class (tensors only), ``ckpt.npz`` its eval-mode
    output on a seeded tile.  The reverse direction is asserted here: state_dicts saved from the new package's
    Generator / Segmentor / Discriminator classes load into the reference classes with strict=True."""
    import io
    torch.manual_seed(1234)
    C = 3
    G = RM.Generator(C)
    with torch.no_grad():                       # make the BN statistics non-trivial, as after training
        for m in G.modules():
            if isinstance(m, nn.BatchNorm2d):
                m.running_mean.normal_(0, 0.1); m.running_var.uniform_(0.6, 1.4)
                m.num_batches_tracked.fill_(17)
    torch.save(G.state_dict(), os.path.join(HERE, 'netG_ref.pkl'))
    G.eval()
    x, _, _ = seeded_tiles(4321, 1, C, 40, 40)
    with torch.no_grad():
        y = G(x)
    np.savez_compressed(os.path.join(HERE, 'ckpt.npz'), out=y.numpy(), meta=np.array([4321, 1, C, 40, 40], np.int64))
    import fcd_gan_pytorch_amd as fcd
    for ours, theirs in ((fcd.Module.Generator(4), RM.Generator(4)),
                         (fcd.Module.Segmentor(4, 1, True), RM.Segmentor(4, 1, True)),
                         (fcd.Module.Segmentor(3, 1, False), RM.Segmentor(3, 1, False)),
                         (fcd.Module.Discriminator_SRGAN_simple(4), RM.Discriminator_SRGAN_simple(4))):
        buf = io.BytesIO()
        torch.save(ours.state_dict(), buf)
        buf.seek(0)
        theirs.load_state_dict(torch.load(buf), strict=True)
        for (k1, v1), (k2, v2) in zip(ours.state_dict().items(), theirs.state_dict().items()):
            assert k1 == k2 and torch.equal(v1, v2), k1
    print('checkpoint interchange: reference-written netG_ref.pkl; ours -> reference strict load OK')


if __name__ == '__main__':
    ap = argparse.ArgumentParser()
    ap.add_argument('--only', default=None)
    a = ap.parse_args()
    torch.manual_seed(0)
    torch.set_num_threads(8)
    RM, RL, RS = load_reference()
    if a.only in (None, 'modules'):
        gen_modules(RM)
    if a.only in (None, 'losses'):
        gen_losses(RL, RS)
    if a.only in (None, 'steps'):
        gen_steps(RM, RL)
    if a.only in (None, 'steps2'):
        gen_steps_extra(RM, RL)
    if a.only in (None, 'ckpt'):
        gen_checkpoint(RM)
