"""GPU parity of the drop-in modules / criteria / train steps.

Two anchors, same seeded inputs and weights:
  (1) the committed golden fixtures produced by the REFERENCE itself
      (tests/golden/*.npz, generator script gen_golden.py);
  (2) the CPU oracle (oracle/) run on the GPU box, for full-tensor comparisons.
Tolerances (fp32, different summation order than oneDNN, through up to ~20 conv
layers with BatchNorm): 1e-4 absolute on sigmoid/density outputs (north_star),
2e-3 relative to the tensor's max on gradients.
"""
import os

import numpy as np
import pytest
import torch

from seeded import seeded_state, seeded_tiles, summary
from oracle import nets as onets, losses as olosses, steps as osteps

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(__file__), 'golden')
DEV = 'cuda'


def pkg():
    import fcd_gan_pytorch_amd as p
    return p


def probe_like(shape, seed):
    rng = np.random.default_rng([991, seed])
    return torch.from_numpy(rng.standard_normal(tuple(shape)).astype(np.float32))


def rel_err(got, ref):
    got, ref = got.detach().cpu().double(), ref.detach().cpu().double()
    return ((got - ref).abs().max() / ref.abs().max().clamp_min(1e-12)).item()


def grad_close(got, ref, what, l2_tol=5e-3, floor=0.0, bad_frac=5e-3):
    """Gradient comparison robust to the non-smooth points of the nets: a ReLU / LeakyReLU
    pre-activation or a max-pool tie within rounding distance of the kink flips a whole
    gradient path (any two fp32 implementations differ there, including the reference with a
    different thread count), which shows up as a FEW locally large errors.  So: relative L2
    error small, and all but 0.5% of the elements within 1e-3 of the tensor's max."""
    got, ref = got.detach().cpu().double(), ref.detach().cpu().double()
    assert got.shape == ref.shape, what
    scale = max(ref.abs().max().item(), floor)
    if scale == 0.0:
        assert got.abs().max().item() == 0.0, what
        return
    d = (got - ref).abs()
    l2 = (d.norm() / max(ref.norm().item(), floor * ref.numel() ** 0.5, 1e-30)).item()
    frac_bad = (d > 1e-3 * scale).double().mean().item()
    assert l2 < l2_tol, '%s: relative L2 error %.3e' % (what, l2)
    assert frac_bad < bad_frac, '%s: %.3f%% of elements off by > 1e-3*max' % (what, 100 * frac_bad)


def is_pre_bn_bias(key):
    """Conv biases that feed a BatchNorm: their true gradient is exactly 0 (BN removes the
    mean), what any implementation computes is rounding noise, and RMSprop/Adam turn that
    noise into +-lr-sized random steps."""
    import re
    return bool(re.search(r'(double_conv\.[03]\.bias|^net\.[258]\.bias|block[2-6]\.conv[12]\.bias|block7\.0\.bias)$', key))


def build(tag, C, w_seed):
    M = pkg().Module
    kind = tag[0]
    if kind == 'G':
        m, spec = M.Generator(C), onets.generator_spec(C)
    elif kind == 'S':
        bil = tag[2] == 'b'
        m, spec = M.Segmentor(C, 1, bil), onets.segmentor_spec(C, 1, bil)
    else:
        m, spec = M.Discriminator_SRGAN_simple(C), onets.discriminator_spec(C)
    # drop-in contract: identical state_dict keys and shapes
    assert [(k, tuple(v.shape)) for k, v in m.state_dict().items()] == [(k, tuple(s)) for k, s in spec.items()]
    sd = seeded_state(spec, w_seed)
    m.load_state_dict(sd)
    return m.to(DEV), sd


MODULE_CASES = ['G4_32', 'G4_32_eval', 'G13_24x40', 'S4b_32', 'S4b_40x56', 'S4b_32_eval', 'S4t_32',
                'S3t_40x56', 'D4_32', 'D4_48x40', 'D3_38x50']


@pytest.mark.parametrize('tag', MODULE_CASES)
def test_module_vs_reference_fixture_and_oracle(tag):
    z = np.load(os.path.join(G, 'modules.npz'))
    tile_seed, w_seed, ci, N, C, H, W, train = [int(v) for v in z[tag + '/meta']]
    x, y, _ = seeded_tiles(tile_seed, N, C, H, W)
    m, sd = build(tag, C, w_seed)
    m.train(bool(train))
    xg, yg = x.to(DEV).requires_grad_(True), y.to(DEV).requires_grad_(True)
    o = m(xg) if tag[0] == 'G' else m(xg, yg)
    pr = probe_like(o.shape, ci)
    (o * pr.to(DEV)).sum().backward()
    # (1) reference fixture: full output
    ref_out = torch.from_numpy(z[tag + '/out'])
    tol = 1e-4 if tag[0] != 'G' else 1e-4 * max(1.0, float(ref_out.abs().max()))
    assert (o.detach().cpu() - ref_out).abs().max().item() <= tol
    # (2) oracle: full gradients + buffers.  Train-mode BN over the tiny per-channel populations
    # of these small cases (down to N*2*2 values) makes the S gradients ill-conditioned: the
    # oracle's OWN gradients move by ~1e-2 (relative L2) under a 1e-6 relative perturbation of
    # the weights.  The tolerance is therefore condition-aware: 4x the oracle's response to a
    # seeded 1e-6 weight perturbation, floored at 5e-3.
    kinks = [0]

    def run_oracle(eps):
        # count the oracle's activation inputs within 2e-5 of the kink (fp32 conv outputs of O(10)
        # carry ~1e-6 of rounding): one of them falling the other way flips a whole gradient path
        import torch.nn.functional as F_
        saved = {n: getattr(F_, n) for n in ('relu', 'leaky_relu', 'prelu')}

        def counting(fn):
            def f(t, *a, **k):
                kinks[0] += int((t.detach().abs() < 2e-5).sum())
                return fn(t, *a, **k)
            return f
        for n_, fn in saved.items():
            setattr(F_, n_, counting(fn))
        try:
            return run_oracle_(eps)
        finally:
            for n_, fn in saved.items():
                setattr(F_, n_, fn)

    def run_oracle_(eps):
        osd = onets.clone_state(sd)
        if eps:
            gen = torch.Generator().manual_seed(1234)
            with torch.no_grad():
                for k in onets.param_keys(osd):
                    osd[k].mul_(1 + eps * torch.randn(osd[k].shape, generator=gen))
        xr, yr = x.clone().requires_grad_(True), y.clone().requires_grad_(True)
        if tag[0] == 'G':
            orf = onets.generator(osd, xr, train=bool(train))
        elif tag[0] == 'S':
            orf = onets.segmentor(osd, xr, yr, train=bool(train), bilinear=tag[2] == 'b')
        else:
            orf = onets.discriminator(osd, xr, yr, train=bool(train))
        (orf * pr).sum().backward()
        return osd, xr.grad, yr.grad

    def rl2(a, b):
        a, b = a.detach().cpu().double(), b.detach().cpu().double()
        return ((a - b).norm() / b.norm().clamp_min(1e-30)).item()

    osd, dxr, dyr = run_oracle(0.0)
    psd, dxp, dyp = run_oracle(1e-6)
    keys = [k for k, _ in m.named_parameters() if not (train and is_pre_bn_bias(k))]
    flat = lambda d: torch.cat([d[k].grad.reshape(-1) for k in keys])
    sens_x = rl2(dxp, dxr)
    sens_w = rl2(flat(psd), flat(osd))
    # kink-aware floor: with activation inputs of the oracle sitting on the kink, a flipped unit
    # (different, equally valid fp32 summation order) moves the gradients by ~1e-2 relative L2
    floor = 3e-2 if kinks[0] > 0 else 5e-3
    tol_x, tol_w = max(floor, 4 * sens_x), max(floor, 4 * sens_w)
    assert rl2(xg.grad, dxr) < tol_x, 'dx rel-L2 %.2e (oracle sensitivity %.2e, %d kink inputs)' % (
        rl2(xg.grad, dxr), sens_x, kinks[0])
    if tag[0] != 'G':
        assert rl2(yg.grad, dyr) < max(floor, 4 * rl2(dyp, dyr)), 'dy'
    gp = dict(m.named_parameters())
    got_w = torch.cat([gp[k].grad.detach().cpu().reshape(-1) for k in keys])
    assert rl2(got_w, flat(osd)) < tol_w, 'param grads rel-L2 %.2e (sens %.2e)' % (rl2(got_w, flat(osd)), sens_w)
    wmax = max(osd[k].grad.abs().max().item() for k in keys if k.endswith('weight'))
    for k, p_ in gp.items():
        if train and is_pre_bn_bias(k):
            assert p_.grad.abs().max().item() < 1e-3 * wmax, k      # analytically zero
        elif osd[k].grad.norm().item() > 1e-4 * wmax:
            assert rl2(p_.grad, osd[k].grad) < 10 * tol_w, 'grad ' + k
    if train:
        for k, v in m.state_dict().items():
            if 'running_' in k:
                assert rel_err(v, osd[k]) < 1e-4, k
            elif 'num_batches' in k:
                assert int(v) == int(osd[k]), k


def test_discriminator_shared_first_argument_equals_two_calls():
    """``forward_shared_first`` (x through D's net once, Demo_RSSS.py:293,302) against the reference's two train-mode calls
    ``netD(x, y1)``, ``netD(x, y2)`` on a second copy of the net: outputs, parameter gradients of ``1 + mean(out2) - mean(out1)``,
    BatchNorm running statistics (the replayed order x, y1, x, y2) and batch counters."""
    a, sd = build('D4_32', 4, 77)
    b, _ = build('D4_32', 4, 77)
    a.train(); b.train()
    n = 3
    x, y1, y2 = (probe_like((n, 4, 64, 48), 20 + i).to(DEV) for i in range(3))
    o1, o2 = a.forward_shared_first(torch.cat([x, y1, y2], 0), 2)
    (1 + o2.mean() - o1.mean()).backward()
    r1 = b(x, y1)
    r2 = b(x, y2)
    (1 + r2.mean() - r1.mean()).backward()
    assert (o1 - r1).abs().max().item() <= 2e-6 and (o2 - r2).abs().max().item() <= 2e-6
    for (k, pa), (_, pb) in zip(a.named_parameters(), b.named_parameters()):
        if is_pre_bn_bias(k):
            continue
        grad_close(pa.grad, pb.grad, 'shared-first ' + k, l2_tol=2e-4, bad_frac=1e-3)
    for (k, ba), (_, bb) in zip(a.named_buffers(), b.named_buffers()):
        if k.endswith('num_batches_tracked'):
            assert int(ba) == int(bb) == 4, k
        else:
            assert rel_err(ba, bb) <= 1e-6, (k, rel_err(ba, bb))
    # eval mode takes the repeated batch (no replay there): same values as two plain calls
    a.eval(); b.eval()
    with torch.no_grad():
        e1, e2 = a.forward_shared_first(torch.cat([x, y1, y2], 0), 2)
        assert (e1 - b(x, y1)).abs().max().item() <= 2e-6 and (e2 - b(x, y2)).abs().max().item() <= 2e-6


def test_binary_map_bit_exact_with_margin():
    """north_star: thresholded change map bit-exact.  Pixels whose density lies within
    the achieved error of the threshold are excluded (SURVEY.md section 7 'hard parts')."""
    z = np.load(os.path.join(G, 'modules.npz'))
    for tag in ('S4b_32', 'S4b_40x56', 'S4b_32_eval'):
        tile_seed, w_seed, ci, N, C, H, W, train = [int(v) for v in z[tag + '/meta']]
        x, y, _ = seeded_tiles(tile_seed, N, C, H, W)
        m, _ = build(tag, C, w_seed)
        m.train(bool(train))
        with torch.no_grad():
            o = m(x.to(DEV), y.to(DEV)).cpu()
        ref = torch.from_numpy(z[tag + '/out'])
        safe = (ref - 0.5).abs() > 1e-4
        assert torch.equal((o > 0.5)[safe], (ref > 0.5)[safe])
        assert safe.float().mean() > 0.99


def make_cmap(seed, N, H, W, all_changed=None):
    rng = np.random.default_rng([555, seed])
    c = torch.from_numpy(rng.uniform(0.02, 0.98, (N, 1, H, W)).astype(np.float32))
    if all_changed is not None:
        c[all_changed] = 1.0
    return c


CRIT = {'cnet_pb': ('CNetLoss', 4, 1, True), 'cnet_rgb2': ('CNetLoss', 3, 2, False),
        'cgen_rgb': ('CGeneratorLoss', 3, 1, False), 'cgen_pb_allchanged': ('CGeneratorLoss', 4, 1, True)}


def make_crit(cls, channel, layer, pb):
    import warnings
    L = pkg().Loss
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        crit = getattr(L, cls)(channel=channel, perception_layer=layer, perception_perBand=pb, allow_seeded=True)
    crit.loss_perception.net.load_state_dict(seeded_state(onets.vgg_spec(), 4242))
    return crit.to(DEV)


@pytest.mark.parametrize('tag', list(CRIT))
def test_criteria_vs_reference_fixture(tag, conv_path):
    z = np.load(os.path.join(G, 'losses.npz'))
    seed, cseed, N, C, H, W, allc = [int(v) for v in z[tag + '/meta']]
    cls, channel, layer, pb = CRIT[tag]
    crit = make_crit(cls, channel, layer, pb)
    t, g, _ = seeded_tiles(seed, N, C, H, W)
    cmap = make_cmap(cseed, N, H, W, None if allc < 0 else allc)
    tg, gg, cg = t.to(DEV), g.to(DEV).requires_grad_(True), cmap.to(DEV).requires_grad_(True)
    vals = crit(tg, gg, cg)
    tot = sum(w * v for w, v in zip([1.0, 0.3, 0.7, 0.2], vals))
    tot.backward()
    np.testing.assert_allclose([float(v) for v in vals], z[tag + '/vals'], rtol=2e-4, atol=1e-6)
    # gradients vs the oracle (full tensors)
    vgg = seeded_state(onets.vgg_spec(), 4242)
    gr, cr = g.clone().requires_grad_(True), cmap.clone().requires_grad_(True)
    if cls == 'CNetLoss':
        rv = olosses.cnet_loss(vgg, t, gr, cr, False, layer, pb)
    else:
        rv = olosses.cgenerator_loss(vgg, t, gr, cr, layer, pb)
    sum(w * v for w, v in zip([1.0, 0.3, 0.7, 0.2], rv)).backward()
    # 13 ReLU layers + 4 max-pools sit between the loss and these gradients: every pre-activation
    # within rounding distance of 0 (and every pool tie) that falls the other way re-routes the
    # gradient of a whole receptive field, so a few % of the elements may differ visibly while the
    # relative L2 error stays small.  (tools/dbg_replay.py replays each conv op in isolation to
    # tell such flips from kernel errors: every op agrees to rounding.)
    l2 = 5e-3 if conv_path == 'direct' else 1e-2
    grad_close(gg.grad, gr.grad, 'd/dgenerated', l2_tol=l2, bad_frac=3e-2)
    grad_close(cg.grad, cr.grad, 'd/dcmap', l2_tol=l2, bad_frac=3e-2)


def test_mse_halves_is_mse_loss_of_the_two_halves():
    """Loss.mse_halves (one autograd node writing both halves of the tap gradient) vs F.mse_loss on the slices in fp64
    (reference Loss.py:57-59: MSE between the target's and the generated image's features)."""
    L = pkg().Loss
    torch.manual_seed(5)
    f = torch.randn(6, 7, 9, 11)
    fr = f.double().requires_grad_(True)
    ref = torch.nn.functional.mse_loss(fr[:3], fr[3:])
    (ref * 1.7).backward()
    fg = f.to(DEV).requires_grad_(True)
    got = L.mse_halves(fg * 1.0, 3)              # a non-leaf input, as in the criterion
    (got * 1.7).backward()
    np.testing.assert_allclose(got.item(), ref.item(), rtol=2e-6)
    np.testing.assert_allclose(fg.grad.cpu().double().numpy(), fr.grad.numpy(), rtol=1e-5, atol=1e-9)
    with pytest.raises(ValueError):
        L.mse_halves(fg, 2)


def test_region_loss_vs_reference_fixture():
    z = np.load(os.path.join(G, 'losses.npz'))
    L = pkg().Loss
    cm = make_cmap(77, 3, 40, 48).to(DEV).requires_grad_(True)
    reg = torch.zeros(3, 1, 40, 48); reg[0, :, 5:20, 8:30] = 1; reg[2, :, 0:40, 0:10] = 1
    reg = reg.to(DEV)
    a = L.region_loss(cm, reg, torch.nn.L1Loss()); b = L.region_loss(cm, 1 - reg, torch.nn.MSELoss())
    (a + 2 * b).backward()
    np.testing.assert_allclose([a.item(), b.item()], z['region/vals'], rtol=1e-5)
    got = summary(cm.grad.cpu())
    np.testing.assert_allclose(got, z['region/dcmap'], rtol=1e-3, atol=1e-9)


def _load_nets(C, wseed, crit_cls, pb, opt_kind):
    p = pkg()
    M, O = p.Module, p.optim
    netG = M.Generator(C); netG.load_state_dict(seeded_state(onets.generator_spec(C), wseed + 1))
    netS = M.Segmentor(C, 1, True); netS.load_state_dict(seeded_state(onets.segmentor_spec(C, 1, True), wseed + 2))
    netD = M.Discriminator_SRGAN_simple(C); netD.load_state_dict(seeded_state(onets.discriminator_spec(C), wseed + 3))
    netG.to(DEV); netS.to(DEV); netD.to(DEV)
    crit = make_crit(crit_cls, C, 1, pb)
    if opt_kind == 'rsss':
        opts = dict(S=O.RMSprop(netS.parameters(), lr=5e-5), D=O.RMSprop(netD.parameters(), lr=5e-5))
    elif opt_kind == 'wsss':
        opts = dict(S=O.RMSprop(netS.parameters(), lr=1e-3), D=O.RMSprop(netD.parameters(), lr=1e-5))
    else:
        opts = dict(S=O.Adam(netS.parameters(), lr=2e-4, betas=(0.9, 0.99)),
                    G=O.Adam(netG.parameters(), lr=2e-4, betas=(0.9, 0.99)))
    return netG, netS, netD, crit, opts


def _update_err(net, before, z, tag, rtol_l2):
    """compare post-step weights with the reference fixture in aggregate: the L2 norm and the
    sum of every tensor (sign-like first optimizer steps make single elements noisy)."""
    for k, v in net.state_dict().items():
        ref = z['%s/%s' % (tag, k)]
        if not v.is_floating_point():
            assert float(v) == float(ref), k
            continue
        if is_pre_bn_bias(k):
            continue
        got = summary(v.cpu())
        assert abs(got[1] - ref[1]) <= rtol_l2 * max(ref[1], 1e-6) + 1e-6, (k, got[1], ref[1])


@pytest.mark.parametrize('literal', [False, True])
def test_rsss_adversarial_step_vs_reference_fixture(literal, conv_path):
    z = np.load(os.path.join(G, 'steps.npz'))
    wseed, tseed, N, C, H, W = [int(v) for v in z['rsss/meta']]
    netG, netS, netD, crit, opts = _load_nets(C, wseed, 'CGeneratorLoss', True, 'rsss')
    netS.train(); netD.train(); netG.eval()
    x, y, region = (t.to(DEV) for t in seeded_tiles(tseed, N, C, H, W))
    steps = pkg().steps
    for it in range(2):
        r = steps.rsss_adversarial_step(netS, netD, netG, crit, opts['S'], opts['D'], x, y, region, literal=literal)
        got = [float(r[k]) for k in ('d_loss', 's_loss', 's_d_loss', 'g_loss', 'l1_loss', 'r_loss',
                                     'generator_loss', 'ssim_loss', 'perception_loss')]
        np.testing.assert_allclose(got, z['rsss/it%d/scalars' % it], rtol=2e-3, atol=1e-5)
        ref_cmap = torch.from_numpy(z['rsss/it%d/cmap' % it])
        # it 1 sees weights moved by sign-like RMSprop steps (first step = +-10 lr whatever |g|): every
        # gradient element at noise level lands on either side, so the bound measures that amplification,
        # not the kernels; the Winograd transforms carry ~10x the rounding of the direct kernels
        tol = 1e-4 if it == 0 else (1e-2 if conv_path == 'direct' else 2e-2)
        assert (r['cmap'].detach().cpu()[:, :, ::4, ::4] - ref_cmap).abs().max().item() <= tol
    _update_err(netS, None, z, 'rsss/S', 2e-3)
    _update_err(netD, None, z, 'rsss/D', 2e-3)
    _update_err(netG, None, z, 'rsss/G', 1e-6)


@pytest.mark.parametrize('literal', [False, True])
def test_usss_joint_step_vs_reference_fixture(literal, conv_path):
    z = np.load(os.path.join(G, 'steps.npz'))
    wseed, tseed, N, C, H, W = [int(v) for v in z['usss/meta']]
    netG, netS, netD, crit, opts = _load_nets(C, wseed, 'CNetLoss', True, 'usss')
    netS.train(); netG.train()
    x, y, _ = (t.to(DEV) for t in seeded_tiles(tseed, N, C, H, W))
    steps = pkg().steps
    for it in range(2):
        r = steps.usss_joint_step(netS, netG, crit, opts['S'], opts['G'], x, y, literal=literal)
        got = [float(r[k]) for k in ('loss', 'net_loss', 'generator_loss', 'l1_loss', 'perception_loss', 'ssim_loss')]
        np.testing.assert_allclose(got, z['usss/it%d/scalars' % it], rtol=3e-3, atol=1e-5)
    _update_err(netS, None, z, 'usss/S', 2e-3)
    _update_err(netG, None, z, 'usss/G', 2e-3)


def test_wsss_adversarial_step_vs_reference_fixture():
    z = np.load(os.path.join(G, 'steps.npz'))
    wseed, tseed, N, C, H, W = [int(v) for v in z['wsss/meta']]
    netG, netS, netD, crit, opts = _load_nets(C, wseed, 'CGeneratorLoss', False, 'wsss')
    netS.train(); netD.train(); netG.eval()
    x, y, _ = (t.to(DEV) for t in seeded_tiles(tseed, N, C, H, W))
    x_nc = seeded_tiles(tseed + 100, N, C, H, W)[0]
    y_nc = (x_nc + 0.05 * seeded_tiles(tseed + 200, N, C, H, W)[0]).to(DEV)
    x_nc = x_nc.to(DEV)
    r = pkg().steps.wsss_adversarial_step(netS, netD, netG, crit, opts['S'], opts['D'], x, y, x_nc, y_nc)
    got = [float(r[k]) for k in ('d_loss', 's_loss', 's_d_loss', 'g_loss', 'l1_loss', 'nc_loss',
                                 'generator_loss', 'ssim_loss', 'perception_loss')]
    np.testing.assert_allclose(got, z['wsss/it0/scalars'], rtol=2e-3, atol=1e-5)
    _update_err(netS, None, z, 'wsss/S', 5e-3)
    _update_err(netD, None, z, 'wsss/D', 2e-3)
