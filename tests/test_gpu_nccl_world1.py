"""RCCL rehearsal on the 1-GPU box: a ONE-rank ``nccl`` process group with
``dp.force_exchange(True)`` sends the data-parallel step through every collective the 8-GPU run will issue --
``sync_start`` / ``sync_buffers`` broadcasts, the bucketed asynchronous ``all_reduce`` calls launched from the
gradient-ready hooks while backward is still running, ``work.wait()`` (a stream-level wait on RCCL's stream), the SyncBN
sums between the split BatchNorm kernels, the epoch reductions -- and must reproduce the no-exchange step BIT FOR BIT
(a one-rank sum is the identity, grad_scale = 1/1).  RCCL refuses two ranks on one device, so this is the only way the
``nccl`` branches execute before the driver's multi-GPU run; the multi-rank logic itself is covered over gloo
(tests/test_dp_gloo.py, tests/test_gpu_dp.py)."""
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


def _step(sync_bn, share=True, head_fused=True):
    import fcd_gan_pytorch_amd as p
    dev = torch.device('cuda', 0)
    p.set_sync_batchnorm(sync_bn)
    # [r5] the Discriminator step sends the shared masked x through D's net once (steps.py, switch D_SHARE); under SyncBN it keeps the
    # reference's two passes.  share=False gives the per-replica run in that same form, so that the SyncBN comparison below
    # measures the SyncBN kernels and not the two ways of summing D's gradient (three RMSprop steps amplify those: D's loss
    # subtracts the two calls, the x-branch gradients nearly cancel)
    p._lib.set_switch('D_SHARE', 1 if share else 0)
    # [r5] likewise the change-density head behind the last BatchNorm and the encoder tails (ops.bn_relu_pool_skip: fp64 channel sums
    # in another order): per replica the head runs inside the head's kernels
    # (ops.bn_relu_head: w[c] * sum instead of sum of w[c] * term in its gradients, ~1e-7 relative), under SyncBN as BatchNorm
    # kernels + head kernels.  head_fused=False gives the per-replica run in the SyncBN run's form (ops.bn_relu_head_ok reads the
    # switch per call; the BatchNorm-in-the-loader fusion of the 3x3 layers is bit-identical either way)
    p._lib.set_switch('BN_FUSE', 1 if head_fused else 0)
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
    p.dp.sync_start((netS, netD, netG), (oS, oD))                    # ncclBroadcast of the flat buffers + BN buffers
    store = {}
    oS.pre_step_hooks.append(lambda o: store.__setitem__('S', (o.flat_g.clone(), o.grad_scale)))
    oD.pre_step_hooks.append(lambda o: store.__setitem__('D', (o.flat_g.clone(), o.grad_scale)))
    x, y, region = (t.to(dev) for t in seeded_tiles(88, N, C, H, H))
    fn = p.steps.rsss_adversarial_step
    for _ in range(3):                                                # later steps: buckets re-armed, hooks re-used
        r = fn(netS, netD, netG, crit, oS, oD, x, y, region)
    p.dp.sync_buffers((netS, netD))
    counts = p.steps.confusion_counts(r['cmap'].detach(), region)     # all-reduced int64 counts
    means = p.dp.mean_scalars(torch.stack([r['s_loss'].detach(), r['d_loss'].detach()]), weight=float(N))
    torch.cuda.synchronize()
    return dict(gS=store['S'][0].cpu().numpy(), gD=store['D'][0].cpu().numpy(), scale=(store['S'][1], store['D'][1]),
                pS=oS.flat_p.cpu().numpy(), pD=oD.flat_p.cpu().numpy(), exS=oS.last_exchange, exD=oD.last_exchange,
                rm=netS.inc.double_conv[1].running_mean.cpu().numpy(), counts=counts.cpu().numpy(), means=means.cpu().numpy(),
                losses=np.array([float(r['s_loss']), float(r['d_loss'])]), syncbn_allreduces=p._ops.SYNC_BN.get('calls', 0))


def _digest(r):
    import hashlib
    return {k: hashlib.sha256(np.ascontiguousarray(r[k]).tobytes()).hexdigest() for k in ('gS', 'gD', 'pS', 'pD', 'rm', 'counts', 'losses')}


def _small(r, plain):
    """What travels back to the test process: digests for the bit-equality checks, small records, and the distances to the
    plain run (the flat gradient / parameter buffers themselves are 2 x 163 MB per variant)."""
    def rel(k):
        a, b = r[k].astype(np.float64), plain[k].astype(np.float64)
        return float(np.linalg.norm(a - b) / np.linalg.norm(b))
    return dict(digest=_digest(r), scale=r['scale'], exS=r['exS'], exD=r['exD'], means=r['means'], losses=r['losses'],
                syncbn_allreduces=r['syncbn_allreduces'], rel={k: rel(k) for k in ('gS', 'gD')}, rm=r['rm'])


def _worker(port, q, variants):
    import fcd_gan_pytorch_amd as p
    torch.cuda.set_device(0)
    out = {}
    plain = _step(False)                                              # no process group at all: the single-GPU product path
    out['plain'] = _small(plain, plain)
    os.environ['MASTER_ADDR'] = '127.0.0.1'; os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('nccl', rank=0, world_size=1, device_id=torch.device('cuda', 0))
    try:
        out['backend'] = dist.get_backend()
        if 'idle' in variants:
            out['idle'] = _small(_step(False), plain)                 # group exists, one rank, nothing forced: must skip the exchange
        p.dp.force_exchange(True)
        if 'forced' in variants:
            out['forced'] = _small(_step(False), plain)               # every collective runs, per-replica BatchNorm
        if 'forced_syncbn' in variants:
            p.dp.force_exchange(False)
            plain4 = _step(False, share=False, head_fused=False)      # per-replica statistics, D step and head in the form SyncBN takes
            p.dp.force_exchange(True)
            out['forced_syncbn'] = _small(_step(True), plain4)        # + the SyncBN sums through ncclAllReduce
            out['forced_syncbn']['rm_ref'] = plain4['rm']
        p.dp.force_exchange(False)
    finally:
        dist.destroy_process_group()
    q.put(out)


def _run_worker(variants):
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    s = socket.socket(); s.bind(('127.0.0.1', 0)); port = s.getsockname()[1]; s.close()
    pr = ctx.Process(target=_worker, args=(port, q, variants))
    pr.start()
    import queue as _queue
    import time
    t0, out = time.time(), None
    while out is None:
        try:
            out = q.get(timeout=5)
        except _queue.Empty:
            assert pr.exitcode in (None, 0), 'the nccl world-1 worker died (exit code %s) -- see its traceback above' % pr.exitcode
            assert time.time() - t0 < 600, 'nccl world-1 step did not finish in 10 minutes (dead-lock in a collective?)'
    pr.join(120)
    assert pr.exitcode == 0
    assert out['backend'] == 'nccl'
    return out


def test_nccl_one_rank_forced_exchange_is_bit_identical():
    out = _run_worker(('idle', 'forced', 'forced_syncbn'))
    plain, idle, forced, fsbn = out['plain'], out['idle'], out['forced'], out['forced_syncbn']
    assert idle['exS'] is None and idle['exD'] is None                # one rank, not forced: no collective was issued
    for r in (forced, fsbn):
        assert r['exS']['buckets'] == 4 and r['exD']['buckets'] == 1, (r['exS'], r['exD'])
        assert r['exS']['launched_during_backward'] >= 1, r['exS']    # buckets left from the hooks, during backward
        assert r['scale'] == (1.0, 1.0)
    print('\n[nccl world 1] S buckets %s, %d launched during backward; D %s' % (forced['exS']['bytes'],
                                                                             forced['exS']['launched_during_backward'], forced['exD']['bytes']))
    assert idle['digest'] == plain['digest']
    assert forced['digest'] == plain['digest'], forced['rel']          # the exchange path changes no bit
    np.testing.assert_allclose(forced['means'], forced['losses'], rtol=1e-6)
    # SyncBN computes the statistics with the split kernels (partial sums -> all-reduce -> apply): same numbers up to the
    # summation order of the fp64 partials
    for k in ('gS', 'gD'):
        print('[nccl world 1] SyncBN through ncclAllReduce vs fused per-replica kernels, %s rel-L2 %.2e' % (k, fsbn['rel'][k]))
        assert fsbn['rel'][k] < 2e-3, (k, fsbn['rel'][k])
    np.testing.assert_allclose(fsbn['rm'], fsbn['rm_ref'], rtol=1e-5, atol=1e-7)
    assert fsbn['syncbn_allreduces'] > forced['syncbn_allreduces'] == 0      # the SyncBN sums really went through the process group
