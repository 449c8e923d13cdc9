"""The drop-in surface used the way an UNCHANGED Demo script uses it.

``install_as_reference_modules()``, then the scripts' own import lines (``from Module import *`` /
``from Loss import *`` / ``from ssim import MS_SSIM``), the reference's constructor lines
(Demo_RSSS.py:137-164, Demo_USSS.py:108-131, Demo_WSSS.py:104-131), STOCK ``torch.optim.RMSprop`` /
``Adam`` on ``.parameters()`` and loop bodies written against Demo_RSSS.py:285-343,
Demo_USSS.py:306-341 and Demo_WSSS.py:249-323: two separate ``netD(...)`` calls, ``retain_graph=True``,
one ``.item()`` per loss, the double backward -- nothing from ``fcd_gan_pytorch_amd.steps`` / ``.optim``.

What this exercises that the step-function tests do not: the product caches filter packs, folded
filters and BatchNorm partial sums keyed on ``tensor._version`` / ``data_ptr()``; here the weights move
under a foreign optimizer (foreach / fused ``step()``), gradients live in stock ``.grad`` tensors
(``zero_grad(set_to_none=True)``), and every loss is synchronised to the host between launches.

Anchors: ``tests/golden/steps.npz`` -- written by the REFERENCE with the same stock optimizers
(gen_golden.py:211-293) -- at the tolerances test_gpu_modules.py uses for the fcd optimizers, on both
conv plans; plus bit-level agreement with the product's own step functions (same kernels, same order).
"""
import os
import sys

import numpy as np
import pytest
import torch

from seeded import seeded_state, seeded_tiles, summary
from oracle import nets as onets

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(__file__), 'golden')


@pytest.fixture
def ref_names(tmp_path, monkeypatch):
    """The namespace a Demo script has after its import block, with this package installed under the
    reference's module names.  The VGG16 weights come the way INTEGRATION.md tells a user without
    network to supply them (FCDGAN_VGG16_WEIGHTS), so the criterion constructor line stays the
    reference's (no ``allow_seeded``)."""
    import fcd_gan_pytorch_amd as pkg
    vgg = {'features.' + k: v for k, v in seeded_state(onets.vgg_spec(), 4242).items()}
    path = str(tmp_path / 'vgg16_seeded.pth')
    torch.save(vgg, path)
    monkeypatch.setenv('FCDGAN_VGG16_WEIGHTS', path)
    saved = {k: sys.modules.get(k) for k in ('Module', 'Loss', 'ssim')}
    pkg.install_as_reference_modules()
    ns = {}
    exec('import torch\nimport torch.nn as nn\nfrom Module import *\nfrom Loss import *\nfrom ssim import MS_SSIM\n', ns)
    assert ns['Segmentor'] is pkg.Module.Segmentor and ns['CGeneratorLoss'] is pkg.Loss.CGeneratorLoss
    assert ns['MS_SSIM'] is pkg.ssim.MS_SSIM and ns['region_loss'] is pkg.Loss.region_loss
    yield ns
    for k, v in saved.items():
        if v is None:
            sys.modules.pop(k, None)
        else:
            sys.modules[k] = v


def _is_pre_bn_bias(key):
    import re
    return bool(re.search(r'(double_conv\.[03]\.bias|^net\.[258]\.bias|block[2-6]\.conv[12]\.bias|block7\.0\.bias)$', key))


def _weights_vs_fixture(net, z, tag, rtol_l2):
    for k, v in net.state_dict().items():
        ref = z['%s/%s' % (tag, k)]
        if not v.is_floating_point():
            assert float(v) == float(ref), k
            continue
        if _is_pre_bn_bias(k):
            continue
        got = summary(v.cpu())
        assert abs(got[1] - ref[1]) <= rtol_l2 * max(ref[1], 1e-6) + 1e-6, (tag, k, got[1], ref[1])


def _load(net, spec, seed):
    net.load_state_dict(seeded_state(spec, seed))
    return net


# ------------------------------------------------------------------------------------------ Demo_RSSS
def _rsss_script(ns, device, x, y, region, iters, C):
    """Set-up + adversarial epoch body of Demo_RSSS, names and order as in the script."""
    torch, nn = ns['torch'], ns['nn']
    Discriminator_SRGAN_simple, Segmentor, Generator = ns['Discriminator_SRGAN_simple'], ns['Segmentor'], ns['Generator']
    CGeneratorLoss, region_loss = ns['CGeneratorLoss'], ns['region_loss']
    learning_rate = 5e-5
    perception_weight, ssim_weight = 0.1, 0
    d_weight, l1_weight, g_weight, r_weight = 1, 0.02, 0.5, 2
    discriminator_continuous = True

    netD = Discriminator_SRGAN_simple(n_channels=C)
    netD.to(device)
    netS = Segmentor(n_channels=C, bilinear=True)
    netS.to(device)
    netG = Generator(n_channels=C)
    netG.to(device)
    _load(netG, onets.generator_spec(C), 7001); _load(netS, onets.segmentor_spec(C, 1, True), 7002)
    _load(netD, onets.discriminator_spec(C), 7003)
    netS.train()
    netG.train()
    netD.train()
    optimizerS = torch.optim.RMSprop(netS.parameters(), lr=learning_rate)
    optimizerD = torch.optim.RMSprop(netD.parameters(), lr=learning_rate)
    g_criterion = CGeneratorLoss(channel=C, perception_layer=1, perception_perBand=True)
    g_criterion.to(device)
    netG.eval()

    total_dataset_size = x.size(0) * iters
    log = []
    aver = dict(d=0.0, s_d=0.0, g=0.0, s=0.0, l1=0.0, r=0.0, gen=0.0, ssim=0.0, perc=0.0)
    for _ in range(iters):
        x = x.to(device)
        y = y.to(device)
        region = region.to(device)

        cmap = netS(x, y)
        if discriminator_continuous == True:      # noqa: E712
            cmask = cmap
        else:
            cmask = (torch.sign(cmap - 0.5) + 1) / 2
        x_mask = x * (1 - cmask.repeat((1, x.size()[1], 1, 1)))
        y_mask = y * (1 - cmask.repeat((1, y.size()[1], 1, 1)))
        c_out = netD(x_mask, y_mask)

        x_unc = x
        y_unc = y * (1 - region) + x * region
        x_unc = x_unc * (1 - cmask.repeat((1, x.size()[1], 1, 1)))
        y_unc = y_unc * (1 - cmask.repeat((1, y.size()[1], 1, 1)))
        nc_out = netD(x_unc, y_unc)

        optimizerD.zero_grad()
        d_loss = 1 + nc_out.mean() - c_out.mean()
        d_loss.backward(retain_graph=True)
        optimizerD.step()

        c_out = netD(x_mask, y_mask)
        y_fake = netG(x)
        generator_loss, ssim_loss, perception_loss = g_criterion(y, y_fake, cmap)
        g_loss = generator_loss + perception_weight * perception_loss + ssim_weight * ssim_loss
        criterion = nn.L1Loss()
        l1_loss = region_loss(cmap, region, criterion)
        s_d_loss = c_out.mean()
        criterion = nn.MSELoss()
        r_loss = region_loss(cmap, 1 - region, criterion)
        s_loss = d_weight * s_d_loss + l1_weight * l1_loss + g_weight * g_loss + r_weight * r_loss

        optimizerS.zero_grad()
        s_loss.backward()
        optimizerS.step()

        k = x.size(0) / total_dataset_size
        aver['d'] += d_loss.item() * k
        aver['s_d'] += s_d_loss.item() * k
        aver['g'] += g_loss.item() * k
        aver['s'] += s_loss.item() * k
        aver['l1'] += l1_loss.item() * k
        aver['r'] += r_loss.item() * k
        aver['gen'] += generator_loss.item() * k
        aver['ssim'] += ssim_loss.item() * k
        aver['perc'] += perception_loss.item() * k
        log.append(([v.item() for v in (d_loss, s_loss, s_d_loss, g_loss, l1_loss, r_loss, generator_loss, ssim_loss,
                                         perception_loss)], cmap.detach().cpu()))
    return netS, netD, netG, log, aver


def test_rsss_loop_as_the_script_writes_it(ref_names, conv_path):
    z = np.load(os.path.join(G, 'steps.npz'))
    wseed, tseed, N, C, H, W = [int(v) for v in z['rsss/meta']]
    assert wseed == 7000
    x, y, region = seeded_tiles(tseed, N, C, H, W)
    device = torch.device('cuda:0')
    netS, netD, netG, log, aver = _rsss_script(ref_names, device, x, y, region, 2, C)
    for it, (scalars, cmap) in enumerate(log):
        np.testing.assert_allclose(scalars, z['rsss/it%d/scalars' % it], rtol=2e-3, atol=1e-5)
        tol = 1e-4 if it == 0 else (1e-2 if conv_path == 'direct' else 2e-2)      # (see test_gpu_modules.py: RMSprop's sign-like steps)
        assert (cmap[:, :, ::4, ::4] - torch.from_numpy(z['rsss/it%d/cmap' % it])).abs().max().item() <= tol
    _weights_vs_fixture(netS, z, 'rsss/S', 2e-3)
    _weights_vs_fixture(netD, z, 'rsss/D', 2e-3)
    _weights_vs_fixture(netG, z, 'rsss/G', 1e-6)          # frozen in this phase: untouched
    np.testing.assert_allclose(aver['d'], np.mean([z['rsss/it%d/scalars' % i][0] for i in range(2)]), rtol=2e-3)
    # every parameter of S and D has a stock .grad tensor (nothing was diverted into an fcd optimizer's flat buffer)
    assert all(p.grad is not None and p.grad.is_cuda for p in list(netS.parameters()) + list(netD.parameters()))
    # the script never freezes G's parameters (netG.eval() only switches its BatchNorm): s_loss.backward() leaves gradients on them
    # that nobody steps on, exactly as in the reference
    assert all(p.grad is not None for p in netG.parameters())


def test_rsss_script_loop_equals_the_fused_step_function(ref_names):
    """Same kernels either way: the script-shaped loop on stock optimizers and ``steps.rsss_adversarial_step(literal=True)``
    on the fcd optimizers must land on the same weights up to the two optimizers' rounding (RMSprop: one addcdiv)."""
    import fcd_gan_pytorch_amd as pkg
    N, C, H, W = 2, 4, 176, 176
    x, y, region = seeded_tiles(7100, N, C, H, W)
    device = torch.device('cuda:0')
    netS, netD, netG, log, _ = _rsss_script(ref_names, device, x, y, region, 2, C)
    M, O = pkg.Module, pkg.optim
    S2 = _load(M.Segmentor(C, 1, True), onets.segmentor_spec(C, 1, True), 7002).to(device).train()
    D2 = _load(M.Discriminator_SRGAN_simple(C), onets.discriminator_spec(C), 7003).to(device).train()
    G2 = _load(M.Generator(C), onets.generator_spec(C), 7001).to(device).eval()
    crit = pkg.Loss.CGeneratorLoss(channel=C, perception_layer=1, perception_perBand=True).to(device)
    oS, oD = O.RMSprop(S2.parameters(), lr=5e-5), O.RMSprop(D2.parameters(), lr=5e-5)
    xd, yd, rd = x.to(device), y.to(device), region.to(device)
    for it in range(2):
        r = pkg.steps.rsss_adversarial_step(S2, D2, G2, crit, oS, oD, xd, yd, rd, literal=True)
        got = [float(r[k]) for k in ('d_loss', 's_loss', 's_d_loss', 'g_loss', 'l1_loss', 'r_loss', 'generator_loss',
                                     'ssim_loss', 'perception_loss')]
        np.testing.assert_allclose(log[it][0], got, rtol=2e-3 if it else 1e-5, atol=1e-6)
        if it == 0:     # before any update the two paths run the same kernels on the same numbers
            assert (log[0][1] - r['cmap'].detach().cpu()).abs().max().item() <= 1e-6
    for (k, a), b in zip(netS.state_dict().items(), S2.state_dict().values()):
        if a.is_floating_point() and not _is_pre_bn_bias(k):
            assert abs(a.norm().item() - b.norm().item()) <= 2e-3 * max(b.norm().item(), 1e-6) + 1e-6, k


# ------------------------------------------------------------------------------------------ Demo_USSS
def test_usss_loop_as_the_script_writes_it(ref_names, conv_path):
    """Demo_USSS.py:306-341: optimizerG.zero_grad() first, ``Loss.backward(retain_graph=True)``, then ``optimizerS.zero_grad()``
    and a SECOND backward through the same graph (G's .grad accumulates both), both Adam steps at the end."""
    ns = ref_names
    torch_, Segmentor, Generator, CNetLoss = ns['torch'], ns['Segmentor'], ns['Generator'], ns['CNetLoss']
    z = np.load(os.path.join(G, 'steps.npz'))
    wseed, tseed, N, C, H, W = [int(v) for v in z['usss/meta']]
    device = torch_.device('cuda:0')
    learning_rate, l1_weight, perception_weight, ssim_weight = 2e-4, 0.65, 0.4, 0

    netS = Segmentor(n_channels=C, bilinear=True)
    netS.to(device)
    netG = Generator(n_channels=C)
    netG.to(device)
    _load(netG, onets.generator_spec(C), wseed + 1); _load(netS, onets.segmentor_spec(C, 1, True), wseed + 2)
    netS.train()
    netG.train()
    optimizerG = torch_.optim.Adam(netG.parameters(), lr=learning_rate, betas=(0.9, 0.99))
    optimizerS = torch_.optim.Adam(netS.parameters(), lr=learning_rate, betas=(0.9, 0.99))
    criterion = CNetLoss(channel=C, perception_layer=1, perception_perBand=True)
    criterion.to(device)

    x, y, _ = seeded_tiles(tseed, N, C, H, W)
    NetLoss_aver = 0
    for it in range(2):
        optimizerG.zero_grad()
        x = x.to(device)
        y = y.to(device)
        y_fake = netG(x)
        cmap = netS(x, y)
        generator_loss, l1_loss, perception_loss, ssim_loss = criterion(y, y_fake, cmap)
        Loss = generator_loss + perception_weight * perception_loss + ssim_weight * ssim_loss
        Loss.backward(retain_graph=True)
        NetLoss = generator_loss + l1_weight * l1_loss + perception_weight * perception_loss + ssim_weight * ssim_loss
        optimizerS.zero_grad()
        NetLoss.backward()
        optimizerG.step()
        optimizerS.step()
        NetLoss_aver += NetLoss * x.size(0) / (2 * N)          # the script accumulates the TENSOR here (Demo_USSS.py:343)
        got = [v.item() for v in (Loss, NetLoss, generator_loss, l1_loss, perception_loss, ssim_loss)]
        np.testing.assert_allclose(got, z['usss/it%d/scalars' % it], rtol=3e-3, atol=1e-5)
    assert np.isfinite(float(NetLoss_aver))
    _weights_vs_fixture(netS, z, 'usss/S', 2e-3)
    _weights_vs_fixture(netG, z, 'usss/G', 2e-3)


# ------------------------------------------------------------------------------------------ Demo_WSSS
def test_wsss_loop_as_the_script_writes_it(ref_names):
    """Demo_WSSS.py:249-323 (changed + unchanged pair, nc_loss, lr 1e-3 / 1e-5) on the default plan."""
    ns = ref_names
    torch_ = ns['torch']
    z = np.load(os.path.join(G, 'steps.npz'))
    wseed, tseed, N, C, H, W = [int(v) for v in z['wsss/meta']]
    device = torch_.device('cuda:0')
    netD = ns['Discriminator_SRGAN_simple'](n_channels=C)
    netD.to(device)
    netS = ns['Segmentor'](n_channels=C, bilinear=True)
    netS.to(device)
    netG = ns['Generator'](n_channels=C)
    netG.to(device)
    _load(netG, onets.generator_spec(C), wseed + 1); _load(netS, onets.segmentor_spec(C, 1, True), wseed + 2)
    _load(netD, onets.discriminator_spec(C), wseed + 3)
    netS.train(); netD.train(); netG.eval()
    optimizerS = torch_.optim.RMSprop(netS.parameters(), lr=1e-3)
    optimizerD = torch_.optim.RMSprop(netD.parameters(), lr=1e-5)
    g_criterion = ns['CGeneratorLoss'](channel=C, perception_layer=1, perception_perBand=False)
    g_criterion.to(device)
    x, y, _ = (t.to(device) for t in seeded_tiles(tseed, N, C, H, W))
    x_nc = seeded_tiles(tseed + 100, N, C, H, W)[0]
    y_nc = (x_nc + 0.05 * seeded_tiles(tseed + 200, N, C, H, W)[0]).to(device)
    x_nc = x_nc.to(device)

    cmap = netS(x, y)
    cmask = cmap
    x_mask = x * (1 - cmask.repeat((1, x.size()[1], 1, 1)))
    y_mask = y * (1 - cmask.repeat((1, y.size()[1], 1, 1)))
    c_out = netD(x_mask, y_mask)
    ncmap = netS(x_nc, y_nc)
    x_mask_nc = x_nc * (1 - cmask.repeat((1, x.size()[1], 1, 1)))
    y_mask_nc = y_nc * (1 - cmask.repeat((1, y.size()[1], 1, 1)))
    nc_out = netD(x_mask_nc, y_mask_nc)
    optimizerD.zero_grad()
    d_loss = 1 + nc_out.mean() - c_out.mean()
    d_loss.backward(retain_graph=True)
    optimizerD.step()
    nc_loss = torch_.mean(torch_.pow(ncmap, 2))
    c_out = netD(x_mask, y_mask)
    y_fake = netG(x)
    generator_loss, ssim_loss, perception_loss = g_criterion(y, y_fake, cmap)
    g_loss = generator_loss + 0.5 * perception_loss + 0 * ssim_loss
    l1_loss = torch_.mean(abs(cmap))
    s_d_loss = c_out.mean()
    s_loss = 1 * s_d_loss + 1.6 * l1_loss + 0.2 * g_loss + 1.5 * nc_loss
    optimizerS.zero_grad()
    s_loss.backward()
    optimizerS.step()
    got = [v.item() for v in (d_loss, s_loss, s_d_loss, g_loss, l1_loss, nc_loss, generator_loss, ssim_loss, perception_loss)]
    np.testing.assert_allclose(got, z['wsss/it0/scalars'], rtol=2e-3, atol=1e-5)
    _weights_vs_fixture(netS, z, 'wsss/S', 5e-3)
    _weights_vs_fixture(netD, z, 'wsss/D', 2e-3)


# ------------------------------------------------------------------------------------------ cache invalidation, directly
@pytest.mark.parametrize('opt', ['rmsprop_foreach', 'rmsprop_single', 'adam_fused', 'no_grad_clamp', 'data_edit'])
def test_foreign_weight_updates_invalidate_every_cached_pack(ref_names, opt, conv_path):
    """After ANY foreign in-place update of the parameters the next forward must see the new weights: train-mode output, data and
    weight gradients, and the eval-mode no_grad output (BatchNorm folded into the filters) of a whole Segmentor all equal those of
    a FRESH module built from the updated state_dict (which has no caches to go stale).  ``data_edit``: an edit through ``p.data``
    (the commented-out WGAN clip of Demo_RSSS.py:309-311) bypasses the version counter that autograd and this package key on --
    INTEGRATION.md section 1 tells the user to call ``invalidate_caches(net)`` after such an edit; that is what is tested."""
    import fcd_gan_pytorch_amd as pkg
    ns = ref_names
    torch_ = ns['torch']
    dev = torch_.device('cuda:0')
    C = 4
    spec = onets.segmentor_spec(C, 1, True)
    net = _load(ns['Segmentor'](n_channels=C, bilinear=True), spec, 311).to(dev).train()
    x, y, _ = (t.to(dev) for t in seeded_tiles(312, 2, C, 64, 64))
    net(x, y).mean().backward()          # populates packs for forward and both gradients
    with torch_.no_grad():
        net.eval(); net(x, y); net.train()      # populates the folded-filter cache
    params = list(net.parameters())
    if opt == 'rmsprop_foreach':
        torch_.optim.RMSprop(params, lr=1e-2, foreach=True).step()
    elif opt == 'rmsprop_single':
        torch_.optim.RMSprop(params, lr=1e-2, foreach=False).step()
    elif opt == 'adam_fused':
        torch_.optim.Adam(params, lr=1e-2, fused=True).step()
    elif opt == 'no_grad_clamp':
        with torch_.no_grad():
            for p in params:
                p.add_(p.grad, alpha=-1e-2).clamp_(-0.05, 0.05)
    else:
        for p in params:
            p.data.add_(p.grad, alpha=-1e-2)
            p.data.clamp_(-1, 1)
        pkg.invalidate_caches(net)
    for p in params:
        p.grad = None
    fresh = ns['Segmentor'](n_channels=C, bilinear=True)
    fresh.load_state_dict({k: v.detach().cpu().clone() for k, v in net.state_dict().items()})
    fresh.to(dev).train()
    xa, xb = x.clone().requires_grad_(True), x.clone().requires_grad_(True)
    a, b = net(xa, y), fresh(xb, y)
    assert torch_.equal(a, b)
    a.square().mean().backward(); b.square().mean().backward()
    assert torch_.equal(xa.grad, xb.grad)
    for (k, p), q in zip(net.named_parameters(), fresh.parameters()):
        assert torch_.equal(p.grad, q.grad), k
    with torch_.no_grad():
        net.eval(); fresh.eval()
        assert torch_.equal(net(x, y), fresh(x, y))
