"""Less-travelled branches of the criteria vs the oracle: every VGG tap (feature_layer=5, so the
fused conv+ReLU+pool kernels must step aside wherever a pre-pool activation is tapped), the
binarised-mask switch of CNetLoss (Loss.py:88-91), odd tile sizes."""
import warnings

import numpy as np
import pytest
import torch

from seeded import seeded_state, seeded_tiles
from oracle import nets as onets, losses as olosses

pytestmark = pytest.mark.gpu
DEV = 'cuda'


def _cmap(seed, N, H, W):
    rng = np.random.default_rng([555, seed])
    return torch.from_numpy(rng.uniform(0.02, 0.98, (N, 1, H, W)).astype(np.float32))


@pytest.mark.parametrize('layers,per_band,switch,size', [(5, False, False, 176), (3, True, True, 176), (1, True, False, 200)])
def test_cnet_loss_variants(layers, per_band, switch, size, conv_path):
    import fcd_gan_pytorch_amd as p
    N, C = 1, 3
    vgg = seeded_state(onets.vgg_spec(), 4242)
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        crit = p.Loss.CNetLoss(channel=C, perception_layer=layers, perception_perBand=per_band, allow_seeded=True)
    crit.loss_perception.net.load_state_dict(vgg)
    crit.to(DEV)
    t, g, _ = seeded_tiles(31 + layers, N, C, size, size)
    cmap = _cmap(layers, N, size, size)
    gg, cg = g.to(DEV).requires_grad_(True), cmap.to(DEV).requires_grad_(True)
    vals = crit(t.to(DEV), gg, cg, generator_mask_switch=switch)
    sum(w * v for w, v in zip([1.0, 0.3, 0.7, 0.2], vals)).backward()
    gr, cr = g.clone().requires_grad_(True), cmap.clone().requires_grad_(True)
    ref = olosses.cnet_loss(vgg, t, gr, cr, switch, layers, per_band)
    sum(w * v for w, v in zip([1.0, 0.3, 0.7, 0.2], ref)).backward()
    np.testing.assert_allclose([float(v) for v in vals], [float(v) for v in ref], rtol=3e-4, atol=1e-6)
    for got, want, what in ((gg.grad, gr.grad, 'dgen'), (cg.grad, cr.grad, 'dcmap')):
        d = got.cpu().double() - want.double()
        assert (d.norm() / want.double().norm().clamp_min(1e-30)).item() < (5e-3 if conv_path == 'direct' else 1e-2), what
