"""Pin the CPU oracle (oracle/) against fixtures produced by the reference itself
(tests/golden/gen_golden.py).  Same ATen ops in the same order => the match is
expected to be bit-exact on the generating machine; a tiny tolerance absorbs
CPU-ISA dispatch differences (oneDNN picks kernels per host CPU)."""
import os

import numpy as np
import pytest
import torch

from seeded import seeded_state, seeded_tiles, summary
from oracle import nets, losses, steps

G = os.path.join(os.path.dirname(__file__), 'golden')
RTOL, ATOL = 2e-4, 2e-5


def close(a, b, rtol=RTOL, atol=ATOL):
    a = np.asarray(a, np.float64); b = np.asarray(b, np.float64)
    np.testing.assert_allclose(a, b, rtol=rtol, atol=atol)


def sum_close(t, ref, rtol=5e-4, samp_atol=0.0):
    """compare a tensor with a stored summary (sum, L2, samples)."""
    got = summary(t)
    scale = max(ref[1], 1e-12)
    n = max(t.numel(), 1)
    assert abs(got[1] - ref[1]) <= rtol * scale + 1e-7 + samp_atol * np.sqrt(n) * 0.1, ('L2', got[1], ref[1])
    assert abs(got[0] - ref[0]) <= rtol * scale * np.sqrt(n) + 1e-6 + samp_atol * np.sqrt(n), ('sum', got[0], ref[0])
    np.testing.assert_allclose(got[2:], ref[2:], rtol=5e-3, atol=rtol * scale / np.sqrt(n) * 10 + 1e-7 + samp_atol)


def probe_like(shape, seed):
    rng = np.random.default_rng([991, seed])
    return torch.from_numpy(rng.standard_normal(tuple(shape)).astype(np.float32))


MODULE_CASES = ['G4_32', 'G4_32_eval', 'G13_24x40', 'S4b_32', 'S4b_40x56', 'S4b_32_eval', 'S4t_32',
                'S3t_40x56', 'D4_32', 'D4_48x40', 'D3_38x50']


def run_oracle_module(tag, z):
    tile_seed, w_seed, ci, N, C, H, W, train = [int(v) for v in z[tag + '/meta']]
    x, y, _ = seeded_tiles(tile_seed, N, C, H, W)
    x.requires_grad_(True); y.requires_grad_(True)
    kind = tag[0]
    if kind == 'G':
        sd = nets.clone_state(seeded_state(nets.generator_spec(C), w_seed))
        o = nets.generator(sd, x, train=bool(train))
    elif kind == 'S':
        bil = tag[2] == 'b'
        sd = nets.clone_state(seeded_state(nets.segmentor_spec(C, 1, bil), w_seed))
        o = nets.segmentor(sd, x, y, train=bool(train), bilinear=bil)
    else:
        sd = nets.clone_state(seeded_state(nets.discriminator_spec(C), w_seed))
        o = nets.discriminator(sd, x, y, train=bool(train))
    (o * probe_like(o.shape, ci)).sum().backward()
    return sd, x, y, o


@pytest.mark.parametrize('tag', MODULE_CASES)
def test_modules_match_reference(tag):
    z = np.load(os.path.join(G, 'modules.npz'))
    sd, x, y, o = run_oracle_module(tag, z)
    close(o.detach().numpy(), z[tag + '/out'])
    sum_close(x.grad, z[tag + '/dx'])
    if tag[0] != 'G':
        sum_close(y.grad, z[tag + '/dy'])
    for k in nets.param_keys(sd):
        sum_close(sd[k].grad, z['%s/grad/%s' % (tag, k)])
    for k in z.files:
        if k.startswith(tag + '/buf/'):
            close(sd[k[len(tag) + 5:]].double().numpy(), z[k])


@pytest.mark.parametrize('tag', ['msssim_176', 'msssim_200x184'])
def test_msssim_matches_reference(tag):
    z = np.load(os.path.join(G, 'losses.npz'))
    seed, N, C, H, W = [int(v) for v in z[tag + '/meta']]
    x, y, _ = seeded_tiles(seed, N, C, H, W)
    x.requires_grad_(True); y.requires_grad_(True)
    v = losses.ms_ssim(x, y, data_range=1.0)
    v.backward()
    close(v.item(), z[tag + '/val'], rtol=1e-5)
    sum_close(x.grad, z[tag + '/dx']); sum_close(y.grad, z[tag + '/dy'])
    close(x.grad[0, 0, ::8, ::8].numpy(), z[tag + '/dx_full'], rtol=1e-3, atol=1e-9)
    s, _ = losses.ssim_level(x.detach(), y.detach(), losses.gauss_window())
    close(s.mean().item(), z[tag + '/ssim_val'], rtol=1e-5)


def make_cmap(seed, N, H, W, all_changed=None):
    rng = np.random.default_rng([555, seed])
    c = torch.from_numpy(rng.uniform(0.02, 0.98, (N, 1, H, W)).astype(np.float32))
    if all_changed is not None:
        c[all_changed] = 1.0
    return c


CRIT = {'cnet_pb': ('cnet', 1, True), 'cnet_rgb2': ('cnet', 2, False),
        'cgen_rgb': ('cgen', 1, False), 'cgen_pb_allchanged': ('cgen', 1, True)}


@pytest.mark.parametrize('tag', list(CRIT))
def test_criteria_match_reference(tag):
    z = np.load(os.path.join(G, 'losses.npz'))
    seed, cseed, N, C, H, W, allc = [int(v) for v in z[tag + '/meta']]
    kind, layer, pb = CRIT[tag]
    vgg = seeded_state(nets.vgg_spec(), 4242)
    t, g, _ = seeded_tiles(seed, N, C, H, W)
    cmap = make_cmap(cseed, N, H, W, None if allc < 0 else allc)
    g.requires_grad_(True); cmap.requires_grad_(True)
    if kind == 'cnet':
        vals = losses.cnet_loss(vgg, t, g, cmap, False, layer, pb)
    else:
        vals = losses.cgenerator_loss(vgg, t, g, cmap, layer, pb)
    tot = sum(w * v for w, v in zip([1.0, 0.3, 0.7, 0.2], vals))
    tot.backward()
    close([float(v) for v in vals], z[tag + '/vals'], rtol=2e-5)
    sum_close(g.grad, z[tag + '/dgen']); sum_close(cmap.grad, z[tag + '/dcmap'])


def test_region_loss_matches_reference():
    z = np.load(os.path.join(G, 'losses.npz'))
    cm = make_cmap(77, 3, 40, 48); cm.requires_grad_(True)
    reg = torch.zeros(3, 1, 40, 48); reg[0, :, 5:20, 8:30] = 1; reg[2, :, 0:40, 0:10] = 1
    a = losses.region_loss(cm, reg, 'l1'); b = losses.region_loss(cm, 1 - reg, 'mse')
    (a + 2 * b).backward()
    close([a.item(), b.item()], z['region/vals'], rtol=1e-6)
    sum_close(cm.grad, z['region/dcmap'])


def _state_close(z, tag, sd, rtol=2e-3, step_atol=0.0):
    """step_atol: RMSprop/Adam's first steps are sign-like (|dw| ~ lr/sqrt(1-alpha) whatever |g| is), so a
    gradient element at rounding-noise level may move the other way: allow a few step sizes per sample."""
    for k, v in sd.items():
        ref = z['%s/%s' % (tag, k)]
        if v.is_floating_point():
            sum_close(v, ref, rtol=rtol, samp_atol=0.0 if 'running_' in k else step_atol)
        else:
            assert float(v) == float(ref)


def test_rsss_step_matches_reference():
    z = np.load(os.path.join(G, 'steps.npz'))
    wseed, tseed, N, C, H, W = [int(v) for v in z['rsss/meta']]
    n = steps.Nets(seeded_state(nets.generator_spec(C), wseed + 1), seeded_state(nets.segmentor_spec(C, 1, True), wseed + 2),
                   seeded_state(nets.discriminator_spec(C), wseed + 3), seeded_state(nets.vgg_spec(), 4242)).make_optimizers('rsss')
    x, y, region = seeded_tiles(tseed, N, C, H, W)
    for it in range(2):
        r = steps.rsss_adversarial_step(n, x, y, region)
        got = [float(r[k]) for k in ('d_loss', 's_loss', 's_d_loss', 'g_loss', 'l1_loss', 'r_loss', 'gen', 'ssim', 'perc')]
        close(got, z['rsss/it%d/scalars' % it], rtol=5e-4, atol=1e-6)
        close(r['cmap'].detach()[:, :, ::4, ::4].numpy(), z['rsss/it%d/cmap' % it], rtol=2e-3, atol=2e-5)
    _state_close(z, 'rsss/S', n.S, step_atol=2 * 2 * 5e-4); _state_close(z, 'rsss/D', n.D, step_atol=2 * 2 * 5e-4)
    _state_close(z, 'rsss/G', n.G)


def test_usss_joint_step_matches_reference():
    z = np.load(os.path.join(G, 'steps.npz'))
    wseed, tseed, N, C, H, W = [int(v) for v in z['usss/meta']]
    n = steps.Nets(seeded_state(nets.generator_spec(C), wseed + 1), seeded_state(nets.segmentor_spec(C, 1, True), wseed + 2),
                   None, seeded_state(nets.vgg_spec(), 4242)).make_optimizers('usss')
    x, y, _ = seeded_tiles(tseed, N, C, H, W)
    for it in range(2):
        r = steps.usss_joint_step(n, x, y)
        got = [float(r[k]) for k in ('loss', 'net_loss', 'gen', 'l1', 'perc', 'ssim')]
        close(got, z['usss/it%d/scalars' % it], rtol=5e-4, atol=1e-6)
        sum_close(r['cmap'], z['usss/it%d/cmap_sum' % it], rtol=2e-3)
    _state_close(z, 'usss/S', n.S, step_atol=2 * 2 * 2e-4); _state_close(z, 'usss/G', n.G, step_atol=2 * 2 * 2e-4)


def test_wsss_step_matches_reference():
    z = np.load(os.path.join(G, 'steps.npz'))
    wseed, tseed, N, C, H, W = [int(v) for v in z['wsss/meta']]
    n = steps.Nets(seeded_state(nets.generator_spec(C), wseed + 1), seeded_state(nets.segmentor_spec(C, 1, True), wseed + 2),
                   seeded_state(nets.discriminator_spec(C), wseed + 3), seeded_state(nets.vgg_spec(), 4242)).make_optimizers('wsss')
    x, y, _ = seeded_tiles(tseed, N, C, H, W)
    x_nc, _, _ = seeded_tiles(tseed + 100, N, C, H, W)
    y_nc = x_nc + 0.05 * seeded_tiles(tseed + 200, N, C, H, W)[0]
    r = steps.wsss_adversarial_step(n, x, y, x_nc, y_nc)
    got = [float(r[k]) for k in ('d_loss', 's_loss', 's_d_loss', 'g_loss', 'l1_loss', 'nc_loss', 'gen', 'ssim', 'perc')]
    close(got, z['wsss/it0/scalars'], rtol=5e-4, atol=1e-6)
    sum_close(r['cmap'], z['wsss/it0/cmap_sum'], rtol=2e-3)
    _state_close(z, 'wsss/S', n.S, step_atol=2 * 1e-2); _state_close(z, 'wsss/D', n.D, step_atol=2 * 1e-4)
