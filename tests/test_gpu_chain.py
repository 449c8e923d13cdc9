"""Runs of frozen 3x3 layers (the VGG16 conv + ReLU pairs between two max-pools, reference Loss.py:25-36) through
``ops.frozen_conv_chain`` -- output transform -> input transform fused, the activation between two layers never
written -- against the layer-by-layer ops on the same kernels: BIT-identical outputs and input gradients, and against
the stock fp64 convolution within the F(4x4) tolerance."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def _ops():
    from fcd_gan_pytorch_amd import _ops as ops
    return ops


def rnd(*shape, seed=0, scale=1.0):
    rng = np.random.default_rng([seed, len(shape)] + list(shape))
    return torch.from_numpy((rng.standard_normal(shape) * scale).astype(np.float32))


def _layers(chans, seed):
    ws, bs = [], []
    for i in range(len(chans) - 1):
        w = rnd(chans[i + 1], chans[i], 3, 3, seed=seed + i, scale=(2.0 / (9 * chans[i])) ** 0.5).cuda()
        b = rnd(chans[i + 1], seed=seed + 50 + i, scale=0.1).cuda()
        ws.append(w)          # plain tensors: requires_grad False = frozen
        bs.append(b)
    return ws, bs


def _layerwise(ops, x, ws, bs, pool):
    z = x
    for i, (w, b) in enumerate(zip(ws, bs)):
        if pool and i == len(ws) - 1:
            z = ops.conv2d_relu_maxpool2(z, w, b)
        else:
            z = ops.conv2d(z, w, b, 1, 1, relu=True)
    return z


# N, channels of the run (input first), H, W, pool
CHAIN_CASES = [
    (3, (64, 128, 128), 64, 64, True),          # conv2 block: 16 tiles wide; the data gradient of the first layer is a 64-row GEMM (own kernel)
    (2, (128, 256, 256, 256), 32, 32, True),    # conv3 block at 8 tiles wide (two tile rows per step)
    (2, (128, 256, 256, 256), 32, 32, False),
    (3, (256, 512, 512), 16, 16, False),        # conv5 geometry: 4 x 4 tiles, whole image in one step
    (2, (128, 128, 128), 128, 128, False),      # 32 tiles wide: the whole row in one 8-wave workgroup
    (1, (128, 128, 256), 32, 64, True),         # non-square, 16 tiles wide, 8 tile rows
    (2, (256, 256, 128, 128), 16, 32, False),   # TW = 8 with TH = 4; shrinking channels
    (1, (128, 128, 128), 32, 256, True),        # 64 tiles wide: four 16-tile bands, the neighbouring tile column of a band recomputed
]


@pytest.mark.parametrize('case', CHAIN_CASES, ids=lambda c: '%dx%s@%dx%d%s' % (c[0], '-'.join(map(str, c[1])), c[2], c[3], 'p' if c[4] else ''))
def test_frozen_chain_is_bit_identical_to_the_layerwise_ops(case):
    ops = _ops()
    N, chans, H, W, pool = case
    ws, bs = _layers(chans, seed=7)
    x0 = rnd(N, chans[0], H, W, seed=3).cuda()
    assert ops.frozen_chain_ok(x0, ws), 'the run does not qualify: the test would compare the fallback with itself'

    xa = x0.clone().requires_grad_(True)
    ya = ops.frozen_conv_chain(xa, ws, bs, pool=pool)
    xb = x0.clone().requires_grad_(True)
    yb = _layerwise(ops, xb, ws, bs, pool)
    assert ya.shape == yb.shape
    assert torch.equal(ya, yb), 'forward differs: max %.3e' % (ya - yb).abs().max().item()

    g = rnd(*ya.shape, seed=11).cuda()
    ya.backward(g)
    yb.backward(g)
    assert torch.equal(xa.grad, xb.grad), 'input gradient differs: max %.3e of %.3e' % (
        (xa.grad - xb.grad).abs().max().item(), xb.grad.abs().max().item())

    # second backward through the same node (retain_graph use of the demos): the saved bits are still valid
    xa.grad = None
    ya2 = ops.frozen_conv_chain(xa, ws, bs, pool=pool)
    ya2.backward(g, retain_graph=True)
    g1 = xa.grad.clone()
    xa.grad = None
    ya2.backward(g)
    assert torch.equal(g1, xa.grad)


def test_frozen_chain_against_fp64():
    ops = _ops()
    N, chans, H, W = 2, (128, 256, 256), 32, 32
    ws, bs = _layers(chans, seed=21)
    x0 = rnd(N, chans[0], H, W, seed=5)
    xa = x0.cuda().requires_grad_(True)
    ya = ops.frozen_conv_chain(xa, ws, bs, pool=True)
    g = rnd(*ya.shape, seed=13)
    ya.backward(g.cuda())
    xr = x0.double().requires_grad_(True)
    z = xr
    for w, b in zip(ws, bs):
        z = F.relu(F.conv2d(z, w.cpu().double(), b.cpu().double(), padding=1))
    z = F.max_pool2d(z, 2)
    z.backward(g.double())
    scale = z.abs().max().item()
    assert (ya.detach().cpu().double() - z.detach()).abs().max().item() <= 5e-5 * scale
    # gradients: a ReLU / max-pool decision of a near-zero / near-tied activation may fall the other way in fp32 -- bound the
    # rms, not the maximum
    gerr = (xa.grad.cpu().double() - xr.grad).pow(2).mean().sqrt().item()
    assert gerr <= 1e-2 * xr.grad.pow(2).mean().sqrt().item(), gerr / xr.grad.pow(2).mean().sqrt().item()


def test_frozen_chain_ok_rejects_what_it_cannot_run():
    ops = _ops()
    ws, _ = _layers((128, 256, 256), seed=1)
    assert not ops.frozen_chain_ok(rnd(1, 128, 30, 32).cuda(), ws)        # H % 4 != 0
    assert not ops.frozen_chain_ok(rnd(1, 128, 24, 24).cuda(), ws)        # 6 tiles wide: no kernel
    assert not ops.frozen_chain_ok(rnd(1, 128, 32, 32).cuda(), ws[:1])    # a single layer is not a run
    wt = [w.clone().requires_grad_(True) for w in ws]
    assert not ops.frozen_chain_ok(rnd(1, 128, 32, 32).cuda(), wt)        # trained filters need their own gradients
    w13, _ = _layers((13, 128, 128), seed=2)
    assert not ops.frozen_chain_ok(rnd(1, 13, 32, 32).cuda(), w13)        # 13 reduction channels: first layer is direct


def test_perception_features_use_the_chain_and_match_layerwise(monkeypatch, switches):
    """PerceptionLoss._features on the chain == the same stack layer by layer (switch WINO_CHAIN=0), values and gradients bit for bit."""
    from fcd_gan_pytorch_amd import Loss
    ops = _ops()
    crit = Loss.PerceptionLoss(feature_layer=1, perception_perBand=True, allow_seeded=True).cuda()
    n, C, H, W = 1, 2, 256, 256            # conv5_x at 16 x 16: every block of the stack has a chain kernel
    t = rnd(n, C, H, W, seed=1).cuda()
    g = rnd(n, C, H, W, seed=2).cuda()
    cm = torch.sigmoid(rnd(n, 1, H, W, seed=3)).cuda()
    calls = []
    orig = ops.frozen_conv_chain
    monkeypatch.setattr(ops, 'frozen_conv_chain', lambda *a, **k: (calls.append(len(a[1])), orig(*a, **k))[1])
    ca = cm.clone().requires_grad_(True)
    la = crit(t, g, ca)
    la.backward()
    assert calls == [2, 3, 3, 3], calls            # conv2_x, conv3_x, conv4_x, conv5_x
    switches('WINO_CHAIN', 0)
    cb = cm.clone().requires_grad_(True)
    lb = crit(t, g, cb)
    lb.backward()
    assert len(calls) == 4
    assert torch.equal(la, lb) and torch.equal(ca.grad, cb.grad)
