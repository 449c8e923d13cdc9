"""Remaining train-step bodies vs the CPU oracle on the same
seeded state: USSS G pre-train (Demo_USSS.py:142-159), USSS S pre-train (:219-228), RSSS G
pre-train (Demo_RSSS.py:190-208); plus the 'config A' shape (G-only step, batch 16,
256x256x4) for finiteness / determinism."""
import warnings

import numpy as np
import pytest
import torch

from seeded import seeded_state, seeded_tiles
from oracle import nets as onets, steps as osteps

pytestmark = pytest.mark.gpu
DEV = 'cuda'


def pkg():
    import fcd_gan_pytorch_amd as p
    return p


def make(C, crit_cls, pb, wseed):
    p = pkg()
    sdG = seeded_state(onets.generator_spec(C), wseed + 1)
    sdS = seeded_state(onets.segmentor_spec(C, 1, True), wseed + 2)
    sdV = seeded_state(onets.vgg_spec(), 4242)
    G = p.Module.Generator(C); G.load_state_dict(sdG); G.to(DEV).train()
    S = p.Module.Segmentor(C, 1, True); S.load_state_dict(sdS); S.to(DEV).train()
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        crit = getattr(p.Loss, crit_cls)(channel=C, perception_layer=1, perception_perBand=pb, allow_seeded=True)
    crit.loss_perception.net.load_state_dict(sdV)
    crit.to(DEV)
    return G, S, crit, (sdG, sdS, sdV)


def flat_rel(net, sd):
    """relative L2 distance between the net's parameters and an oracle state_dict, ignoring
    conv biases that feed a BatchNorm (their gradient is rounding noise)."""
    import re
    num = den = 0.0
    for k, v in net.named_parameters():
        if re.search(r'(double_conv\.[03]\.bias|block[2-6]\.conv[12]\.bias|block7\.0\.bias)$', k):
            continue
        d = v.detach().cpu().double() - sd[k].detach().double()
        num += float((d * d).sum()); den += float((sd[k].detach().double() ** 2).sum())
    return (num / den) ** 0.5


def test_usss_pretrain_steps_vs_oracle(conv_path):
    p = pkg()
    C, N, H = 4, 1, 176
    G, S, crit, (sdG, sdS, sdV) = make(C, 'CNetLoss', True, 500)
    oG = p.optim.Adam(G.parameters(), lr=2e-4, betas=(0.9, 0.99))
    oS = p.optim.Adam(S.parameters(), lr=2e-4, betas=(0.9, 0.99))
    n = osteps.Nets(sdG, sdS, None, sdV).make_optimizers('usss')
    x, y, _ = seeded_tiles(501, N, C, H, H)
    xg, yg = x.to(DEV), y.to(DEV)
    r = p.steps.usss_g_pretrain_step(G, crit, oG, xg, yg)
    ro = osteps.usss_g_pretrain_step(n, x, y)
    np.testing.assert_allclose([float(r['loss']), float(r['generator_loss']), float(r['perception_loss'])],
                               [float(ro['loss']), float(ro['gen']), float(ro['perc'])], rtol=2e-3)
    r = p.steps.usss_s_pretrain_step(S, G, crit, oS, xg, yg)
    ro = osteps.usss_s_pretrain_step(n, x, y)
    np.testing.assert_allclose([float(r['net_loss']), float(r['l1_loss']), float(r['perception_loss'])],
                               [float(ro['net_loss']), float(ro['l1']), float(ro['perc'])], rtol=3e-3)
    assert (r['cmap'].detach().cpu() - ro['cmap'].detach()).abs().max().item() < 1e-3   # after one Adam step of G
    wtol = 2e-3 if conv_path == 'direct' else 4e-3
    assert flat_rel(G, n.G) < wtol and flat_rel(S, n.S) < wtol


def test_rsss_g_pretrain_step_vs_oracle():
    p = pkg()
    C, N, H = 4, 2, 176
    G, S, crit, (sdG, sdS, sdV) = make(C, 'CGeneratorLoss', True, 600)
    oG = p.optim.Adam(G.parameters(), lr=5e-5, betas=(0.9, 0.99))
    n = osteps.Nets(sdG, None, None, sdV).make_optimizers_g_only() if hasattr(osteps.Nets, 'make_optimizers_g_only') else None
    if n is None:
        n = osteps.Nets(sdG, None, None, sdV)
        n.opt['G'] = torch.optim.Adam(n.params('G'), lr=5e-5, betas=(0.9, 0.99))
    x, y, region = seeded_tiles(601, N, C, H, H)
    for it in range(2):
        r = p.steps.rsss_g_pretrain_step(G, crit, oG, x.to(DEV), y.to(DEV), region.to(DEV))
        ro = osteps.rsss_g_pretrain_step(n, x, y, region)
        np.testing.assert_allclose([float(r['g_loss']), float(r['generator_loss']), float(r['perception_loss']),
                                    float(r['ssim_loss'])],
                                   [float(ro['g_loss']), float(ro['gen']), float(ro['perc']), float(ro['ssim'])],
                                   rtol=3e-3, atol=1e-5)
    assert flat_rel(G, n.G) < 2e-3


def test_config_a_generator_step_batch16_256():
    """BASELINE.json configs[1]: generator-only fwd/bwd, batch 16, 256x256x4."""
    p = pkg()
    C, N, H = 4, 16, 256
    x, y, _ = (t.to(DEV) for t in seeded_tiles(7, N, C, H, H))

    def run():
        torch.manual_seed(0)
        G = p.Module.Generator(C).to(DEV).train()
        with warnings.catch_warnings():
            warnings.simplefilter('ignore')
            crit = p.Loss.CNetLoss(channel=C, perception_layer=1, perception_perBand=True, allow_seeded=True).to(DEV)
        oG = p.optim.Adam(G.parameters(), lr=2e-4, betas=(0.9, 0.99))
        vals = []
        for _ in range(2):
            r = p.steps.usss_g_pretrain_step(G, crit, oG, x, y, ssim_weight=0.1)
            vals.append([float(r['loss']), float(r['ssim_loss'])])
        return np.array(vals), oG.flat_p.clone()
    a, pa = run()
    b, pb = run()
    assert np.isfinite(a).all() and a[1, 0] < a[0, 0]          # the step reduces the loss
    np.testing.assert_array_equal(a, b)
    assert torch.equal(pa, pb)
