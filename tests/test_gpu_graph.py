"""Train steps replayed from a hipGraph (fcd_gan_pytorch_amd/graph.py) reproduce the launch-by-launch steps BIT FOR BIT:
weights, optimizer state, BatchNorm running statistics and losses after several iterations, with a learning-rate change in
between (the scalars reach the replayed update kernels through device memory) and with Adam's step-dependent bias
correction (Demo_USSS.py:121-122)."""
import warnings

import numpy as np
import pytest
import torch

from seeded import seeded_state, seeded_tiles
from oracle import nets as onets

pytestmark = pytest.mark.gpu
DEV = 'cuda'
C, N, H = 4, 2, 176


def _setup(p):
    netG, netS, netD = p.Module.Generator(C), p.Module.Segmentor(C, 1, True), p.Module.Discriminator_SRGAN_simple(C)
    netG.load_state_dict(seeded_state(onets.generator_spec(C), 101))
    netS.load_state_dict(seeded_state(onets.segmentor_spec(C, 1, True), 102))
    netD.load_state_dict(seeded_state(onets.discriminator_spec(C), 103))
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        crit = p.Loss.CGeneratorLoss(channel=C, perception_layer=1, perception_perBand=True, allow_seeded=True)
    crit.loss_perception.net.load_state_dict(seeded_state(onets.vgg_spec(), 4242))
    for m in (netG, netS, netD, crit):
        m.to(DEV)
    return netG, netS, netD, crit


def _run_rsss(graphed, iters=6, kw_region=False):
    import fcd_gan_pytorch_amd as p
    netG, netS, netD, crit = _setup(p)
    netS.train(); netD.train(); netG.eval()
    oS, oD = p.optim.RMSprop(netS.parameters(), lr=5e-5), p.optim.RMSprop(netD.parameters(), lr=5e-5)
    fn = p.steps.rsss_adversarial_step
    gs = p.graph.GraphedStep(fn, nets=(netS, netD, netG, crit), optimizers=(oS, oD), warmup=2) if graphed else None
    losses = []
    for it in range(iters):
        if it == 4:                                          # a schedule step (CommonFunc.py:23-37 writes param_groups[i]['lr'])
            for o in (oS, oD):
                o.param_groups[0]['lr'] = 2e-5
        x, y, region = (t.to(DEV) for t in seeded_tiles(500 + it, N, C, H, H))      # fresh tiles every iteration
        if kw_region:       # a tensor passed BY KEYWORD is an input like any other (it used to be captured by address)
            r = gs(netS, netD, netG, crit, oS, oD, x, y, region=region) if graphed else fn(netS, netD, netG, crit, oS, oD, x, y, region=region)
            del region          # the caller drops it: a replay must not read the first call's tensor
        else:
            r = gs(netS, netD, netG, crit, oS, oD, x, y, region) if graphed else fn(netS, netD, netG, crit, oS, oD, x, y, region)
        losses.append([float(r[k]) for k in ('d_loss', 's_loss', 'g_loss', 'perception_loss', 'ssim_loss')])
    torch.cuda.synchronize()
    out = dict(pS=oS.flat_p.cpu().numpy(), pD=oD.flat_p.cpu().numpy(), sqS=oS.square_avg.cpu().numpy(), losses=np.array(losses),
               bn={k: v.cpu().numpy() for k, v in list(netS.state_dict().items()) + list(netD.state_dict().items()) if 'running' in k or 'num_batches' in k},
               steps=(oS.steps, oD.steps), versions=netS.inc.double_conv[0].weight._version)
    if graphed:
        out['replays'], out['eager'] = gs.replays, gs.eager_calls
    # the nets stay usable launch by launch after graph replays (packed-filter caches were invalidated, versions bumped)
    netS.eval()
    with torch.no_grad():
        x, y, _ = (t.to(DEV) for t in seeded_tiles(77, 1, C, H, H))
        out['infer'] = netS(x, y).cpu().numpy()
    return out


def test_graphed_rsss_step_equals_eager_steps():
    a, b = _run_rsss(False), _run_rsss(True)
    assert b['replays'] == 4 and b['eager'] == 2
    assert a['steps'] == b['steps'] == (6, 6) and a['versions'] == b['versions']
    np.testing.assert_array_equal(a['losses'], b['losses'])
    for k in ('pS', 'pD', 'sqS', 'infer'):
        np.testing.assert_array_equal(a[k], b[k], err_msg=k)
    for k in a['bn']:
        np.testing.assert_array_equal(a['bn'][k], b['bn'][k], err_msg=k)


def test_graphed_step_refreshes_tensor_keyword_arguments():
    a, b = _run_rsss(False, iters=5, kw_region=True), _run_rsss(True, iters=5, kw_region=True)
    assert b['replays'] == 3 and b['eager'] == 2
    np.testing.assert_array_equal(a['losses'], b['losses'])
    for k in ('pS', 'pD'):
        np.testing.assert_array_equal(a[k], b[k], err_msg=k)


def test_graphed_step_refuses_two_steps_of_one_optimizer():
    """One set of device-side optimizer scalars per replay: a step function that steps the same optimizer twice is run launch by
    launch, with the same results as without the wrapper."""
    import fcd_gan_pytorch_amd as p

    def run(graphed):
        netG, _, _, _ = _setup(p)
        netG.train()
        opt = p.optim.Adam(netG.parameters(), lr=1e-4, betas=(0.9, 0.99))

        def fn(net, o, x, y):
            for _ in range(2):
                o.zero_grad()
                loss = ((net(x) - y) ** 2).mean()
                loss.backward()
                o.step()
            return dict(loss=loss)
        gs = p.graph.GraphedStep(fn, nets=(netG,), optimizers=(opt,), warmup=1) if graphed else None
        for it in range(3):
            x, y, _ = (t.to(DEV) for t in seeded_tiles(900 + it, N, C, 64, 64))
            r = gs(netG, opt, x, y) if graphed else fn(netG, opt, x, y)
        torch.cuda.synchronize()
        return opt.flat_p.cpu().numpy(), float(r['loss']), opt.steps, gs
    pa, la, sa, _ = run(False)
    pb, lb, sb, gs = run(True)
    assert gs.refused_multistep and gs.replays == 0 and not gs.enabled
    assert sa == sb == 6 and la == lb
    np.testing.assert_array_equal(pa, pb)


def _run_usss_adam(graphed, iters=5):
    import fcd_gan_pytorch_amd as p
    netG, _, _, _ = _setup(p)
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        crit = p.Loss.CNetLoss(channel=C, perception_layer=1, perception_perBand=True, allow_seeded=True)
    crit.loss_perception.net.load_state_dict(seeded_state(onets.vgg_spec(), 4242))
    crit.to(DEV)
    netG.train()
    oG = p.optim.Adam(netG.parameters(), lr=1e-4, betas=(0.9, 0.99))
    fn = p.steps.usss_g_pretrain_step
    gs = p.graph.GraphedStep(fn, nets=(netG, crit), optimizers=(oG,), warmup=1) if graphed else None
    vals = []
    for it in range(iters):
        p.optim.adjust_learning_rate(oG, it, lr_start=1e-5, lr_max=3e-4, lr_warm_up_epoch=10)      # Demo_USSS.py:127: a new lr every epoch
        x, y, _ = (t.to(DEV) for t in seeded_tiles(600 + it, N, C, H, H))
        r = gs(netG, crit, oG, x, y) if graphed else fn(netG, crit, oG, x, y)
        vals.append(float(r['loss']))
    torch.cuda.synchronize()
    return dict(p=oG.flat_p.cpu().numpy(), m=oG.exp_avg.cpu().numpy(), v=oG.exp_avg_sq.cpu().numpy(), vals=np.array(vals),
                rm=netG.block2.bn1.running_mean.cpu().numpy(), steps=oG.steps)


def test_graphed_adam_step_with_lr_schedule_equals_eager_steps():
    a, b = _run_usss_adam(False), _run_usss_adam(True)
    assert a['steps'] == b['steps'] == 5
    for k in ('vals', 'p', 'm', 'v', 'rm'):
        np.testing.assert_array_equal(a[k], b[k], err_msg=k)


def test_device_hyper_kernels_equal_scalar_kernels():
    """fcd_adam_step_h / fcd_rmsprop_step_h (scalars from device memory) against the scalar-argument kernels over several steps."""
    import fcd_gan_pytorch_amd as p
    rng = np.random.default_rng(5)
    w0 = torch.from_numpy(rng.standard_normal(10007).astype(np.float32))
    res = {}
    for mode in ('scalar', 'device'):
        for kind in ('adam', 'rmsprop'):
            prm = torch.nn.Parameter(w0.clone().to(DEV))
            opt = p.optim.Adam([prm], lr=3e-4, betas=(0.9, 0.99)) if kind == 'adam' else p.optim.RMSprop([prm], lr=5e-5)
            if mode == 'device':
                opt.use_device_hyper()
            for it in range(7):
                opt.zero_grad()
                opt.param_groups[0]['lr'] = 3e-4 * (1 + it)
                g = torch.from_numpy(np.random.default_rng(it).standard_normal(10007).astype(np.float32)).to(DEV)
                opt.flat_g.copy_(g)
                opt.step()
            res[mode, kind] = opt.flat_p.cpu().numpy()
    for kind in ('adam', 'rmsprop'):
        np.testing.assert_array_equal(res['scalar', kind], res['device', kind], err_msg=kind)
