"""The data-parallel train step on the GPU: two ranks x half batch of the Demo_RSSS adversarial
iteration -- bucketed all-reduce issued from gradient-ready hooks during backward, SyncBN on, 1/world folded into the
update kernel -- reproduce the single-process full-batch gradients and post-step weights.

Transport: ``nccl`` (= RCCL over xGMI) when the box has >= 2 GPUs, one rank per GPU; on a single-GPU box both ranks
share the device and the collectives travel over ``gloo`` (RCCL refuses two ranks on one device), which still runs the
whole product path -- hooks, buckets, async work handles, stream waits, SyncBN exchanges -- on device tensors."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from seeded import seeded_state, seeded_tiles
from oracle import nets as onets

pytestmark = pytest.mark.gpu
C, N, H = 4, 2, 176


def _free_port():
    s = socket.socket(); s.bind(('127.0.0.1', 0)); p = s.getsockname()[1]; s.close()
    return p


def _run_step(x, y, region, sync_bn):
    import fcd_gan_pytorch_amd as p
    dev = torch.device('cuda', torch.cuda.current_device())
    p.set_sync_batchnorm(sync_bn)
    netG, netS, netD = p.Module.Generator(C), p.Module.Segmentor(C, 1, True), p.Module.Discriminator_SRGAN_simple(C)
    netG.load_state_dict(seeded_state(onets.generator_spec(C), 101))
    netS.load_state_dict(seeded_state(onets.segmentor_spec(C, 1, True), 102))
    netD.load_state_dict(seeded_state(onets.discriminator_spec(C), 103))
    crit = p.Loss.CGeneratorLoss(channel=C, perception_layer=1, perception_perBand=True, allow_seeded=True)
    crit.loss_perception.net.load_state_dict(seeded_state(onets.vgg_spec(), 4242))
    for m in (netG, netS, netD, crit):
        m.to(dev)
    netS.train(); netD.train(); netG.eval()
    oS, oD = p.optim.RMSprop(netS.parameters(), lr=5e-5), p.optim.RMSprop(netD.parameters(), lr=5e-5)
    p.dp.sync_start((netS, netD, netG), (oS, oD))
    store = {}
    oS.pre_step_hooks.append(lambda o: store.__setitem__('S', (o.flat_g * o.grad_scale).cpu()))
    oD.pre_step_hooks.append(lambda o: store.__setitem__('D', (o.flat_g * o.grad_scale).cpu()))
    r = p.steps.rsss_adversarial_step(netS, netD, netG, crit, oS, oD, x.to(dev), y.to(dev), region.to(dev))
    torch.cuda.synchronize()
    return dict(gS=store['S'].numpy(), gD=store['D'].numpy(), pS=oS.flat_p.cpu().numpy(), pD=oD.flat_p.cpu().numpy(),
                s_loss=float(r['s_loss']), d_loss=float(r['d_loss']), exS=oS.last_exchange, exD=oD.last_exchange,
                rm=netS.inc.double_conv[1].running_mean.cpu().numpy())


def _worker(rank, world, port, backend, q):
    os.environ['MASTER_ADDR'] = '127.0.0.1'; os.environ['MASTER_PORT'] = str(port)
    torch.cuda.set_device(rank if backend == 'nccl' else 0)
    if backend == 'nccl':
        dist.init_process_group('nccl', rank=rank, world_size=world, device_id=torch.device('cuda', rank))
    else:
        dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        x, y, region = seeded_tiles(88, N, C, H, H)
        sl = slice(rank * N // world, (rank + 1) * N // world)
        q.put((rank, _run_step(x[sl], y[sl], region[sl], True)))
    finally:
        dist.destroy_process_group()


def test_two_rank_step_equals_full_batch():
    backend = 'nccl' if torch.cuda.device_count() >= 2 else 'gloo'
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, backend, q)) for r in range(2)]
    for pr in procs:
        pr.start()
    import queue as _queue
    import time
    res, t0 = {}, time.time()
    while len(res) < 2:
        try:
            rank, out = q.get(timeout=5)
            res[rank] = out
        except _queue.Empty:
            dead = [pr.exitcode for pr in procs if pr.exitcode not in (None, 0)]
            assert not dead, 'a rank died (exit codes %s) -- see its traceback above' % dead
            assert time.time() - t0 < 600, 'two-rank step did not finish in 10 minutes'
    for pr in procs:
        pr.join(120)
        assert pr.exitcode == 0
    x, y, region = seeded_tiles(88, N, C, H, H)
    full = _run_step(x, y, region, False)

    def rl2(a, b):
        return float(np.linalg.norm(a.astype(np.float64) - b) / max(np.linalg.norm(b.astype(np.float64)), 1e-30))
    for rank in (0, 1):
        r = res[rank]
        assert r['exS']['buckets'] == 4 and r['exD']['buckets'] == 1
        assert r['exS']['launched_during_backward'] >= 1, r['exS']
        # the mean loss over ranks is the full-batch loss; gradients after the exchange are the full-batch gradients
        eS, eD = rl2(r['gS'], full['gS']), rl2(r['gD'], full['gD'])
        print('\n[dp %s rank %d] rel-L2 S %.2e  D %.2e  buckets-during-backward %d' % (backend, rank, eS, eD,
                                                                                   r['exS']['launched_during_backward']))
        assert eS <= 5e-3 and eD <= 5e-3, (eS, eD)
        np.testing.assert_allclose(r['rm'], full['rm'], rtol=1e-4, atol=1e-6)       # SyncBN: full-batch running stats
    np.testing.assert_array_equal(res[0]['gS'], res[1]['gS'])                       # both ranks stepped on the same numbers
    np.testing.assert_array_equal(res[0]['pS'], res[1]['pS'])
    np.testing.assert_array_equal(res[0]['pD'], res[1]['pD'])
    np.testing.assert_allclose(0.5 * (res[0]['s_loss'] + res[1]['s_loss']), full['s_loss'], rtol=2e-3)
