"""Data-parallel host logic on CPU (gloo, world_size 2): the flat-buffer optimizer's single
all-reduce + 1/world scaling reproduces the full-batch gradient, and the on-device confusion
counts reduce across ranks.  Compute here is the CPU oracle driven through the PRODUCT's
parameter objects (the HIP kernels need a GPU); what is under test is optim.py / steps.py."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from seeded import seeded_state, seeded_tiles
from oracle import nets as onets


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _net_and_state(C):
    from fcd_gan_pytorch_amd import Module
    g = Module.Generator(C)
    g.load_state_dict(seeded_state(onets.generator_spec(C), 321))
    g.eval()                       # eval-mode BN => the loss is a plain mean over samples
    return g


def _loss(sd, x, y):
    out = onets.generator(sd, x, train=False)
    return ((out - y) ** 2).mean()


def _worker(rank, world, port, q):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        from fcd_gan_pytorch_amd import optim, steps, _lib
        torch.set_num_threads(2)
        C, N = 4, 4
        g = _net_and_state(C)
        opt = optim.Adam(g.parameters(), lr=1e-3, betas=(0.9, 0.99))
        assert all(p.grad.data_ptr() >= opt.flat_g.data_ptr() for p in g.parameters())
        x, y, _ = seeded_tiles(77, N, C, 24, 24)
        shard = slice(rank * N // world, (rank + 1) * N // world)
        sd = dict(g.named_parameters())
        sd.update(dict(g.named_buffers()))
        opt.zero_grad()
        _loss(sd, x[shard], y[shard]).backward()
        local = opt.gather_grads().clone()
        opt.allreduce_grads()
        assert abs(opt.grad_scale - 1.0 / world) < 1e-12
        avg = opt.flat_g * opt.grad_scale
        # step() is a HIP kernel: must refuse CPU tensors loudly
        try:
            opt.step()
            refused = False
        except _lib.FcdError:
            refused = True
        cm = torch.tensor([[0.2, 0.7], [0.9, 0.4]]).view(1, 1, 2, 2) if rank == 0 else \
            torch.tensor([[0.8, 0.1], [0.6, 0.3]]).view(1, 1, 2, 2)
        ref = torch.tensor([[0., 1.], [1., 1.]]).view(1, 1, 2, 2)
        counts = steps.confusion_counts(cm, ref)
        q.put((rank, local.numpy(), avg.numpy(), refused, counts.numpy()))
    finally:
        dist.destroy_process_group()


def test_flat_allreduce_matches_full_batch_gradient():
    world = 2
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=300) for _ in range(world)], key=lambda t: t[0])
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    # single-process full-batch reference through the same product parameter objects
    from fcd_gan_pytorch_amd import optim
    C, N = 4, 4
    g = _net_and_state(C)
    opt = optim.Adam(g.parameters(), lr=1e-3, betas=(0.9, 0.99))
    x, y, _ = seeded_tiles(77, N, C, 24, 24)
    sd = dict(g.named_parameters())
    sd.update(dict(g.named_buffers()))
    opt.zero_grad()
    _loss(sd, x, y).backward()
    full = opt.gather_grads().numpy()
    for rank, local, avg, refused, counts in res:
        assert refused, 'optimizer.step() must refuse CPU tensors'
        np.testing.assert_allclose(avg, full, rtol=1e-4, atol=1e-5 * np.abs(full).max())
        np.testing.assert_array_equal(counts, res[0][4])
    np.testing.assert_allclose(0.5 * (res[0][1] + res[1][1]), full, rtol=1e-4, atol=1e-5 * np.abs(full).max())
    # [tn, fp, fn, tp] over both ranks' 2x2 maps: rank0 pred=[[0,1],[1,0]], rank1 pred=[[1,0],[1,0]], ref=[[0,1],[1,1]]
    np.testing.assert_array_equal(res[0][4], np.array([1, 1, 3, 3]))


def test_lr_schedule_matches_reference_formula():
    from fcd_gan_pytorch_amd.optim import adjust_learning_rate

    class O:
        param_groups = [{'lr': 0.0}]
    o = O()
    # CommonFunc.py:23-37 evaluated by hand
    assert abs(adjust_learning_rate(o, 0, lr_start=1e-4, lr_max=1e-3, lr_warm_up_epoch=5) - 1e-4) < 1e-12
    assert abs(adjust_learning_rate(o, 3, lr_start=1e-4, lr_max=1e-3, lr_warm_up_epoch=5) - (9e-4 / 5 * 3 + 1e-4)) < 1e-12
    assert abs(adjust_learning_rate(o, 12, lr_start=1e-5, lr_max=3e-4, lr_warm_up_epoch=10, lr_sustain_epochs=10) - 3e-4) < 1e-12
    v = adjust_learning_rate(o, 7, lr_start=1e-4, lr_max=1e-3, lr_warm_up_epoch=5)
    assert abs(v - ((1e-3 - 1e-6) * 0.8 ** 2 + 1e-6)) < 1e-12 and o.param_groups[0]['lr'] == v


# ------------------------------------------------------------------ sampler / ragged batches
def test_rank_strided_batches_partition_every_epoch():
    from fcd_gan_pytorch_amd.dp import RankStridedBatches
    n, bs = 23, 3
    one = RankStridedBatches(n, bs, seed=9, rank=0, world=1)
    flat1 = [i for b in one for i in b]
    assert sorted(flat1) == list(range(n)) and [len(b) for b in one] == [3] * 7 + [2]      # no drop_last (Demo_RSSS.py:242)
    assert one.pads == [0] * 8 and len(one) == 8
    for world in (2, 4, 8):
        per_rank = [RankStridedBatches(n, bs, seed=9, rank=r, world=world) for r in range(world)]
        plans = [list(s) for s in per_rank]
        steps_ = {len(p) for p in plans}
        assert len(steps_) == 1, 'every rank must run the same number of steps'
        real = []
        for b in range(len(plans[0])):
            sizes = {len(plans[r][b]) for r in range(world)}
            assert len(sizes) == 1, 'equal local batches (mean of local gradients == global-batch gradient)'
            for r in range(world):
                k = len(plans[r][b]) - per_rank[r].pads[b]
                real += plans[r][b][:k]
        assert sorted(real) == list(range(n)), 'every tile exactly once per epoch, padding excluded'
        # same permutation on every rank: the global batch b is the b-th slice of the world-1 order
        gb = bs * world
        if gb <= n:
            first = sorted(i for r in range(world) for i in plans[r][0])
            assert first == sorted(flat1[:gb])
        drop = [list(RankStridedBatches(n, bs, seed=9, rank=r, world=world, ragged='drop')) for r in range(world)]
        assert all(len(d) == n // gb for d in drop)
    a = list(RankStridedBatches(n, bs, seed=1, rank=0, world=2))
    s = RankStridedBatches(n, bs, seed=0, rank=0, world=2)
    s.set_epoch(1)
    assert list(s) == a and list(RankStridedBatches(n, bs, seed=2, rank=0, world=2)) != a


def test_weighted_ragged_last_batch_reproduces_the_short_batch_gradient():
    """ragged='weighted': the last global batch is dealt out without duplicates; with the per-rank loss scale
    n_local * world / L (steps._backward) the rank-averaged gradient of batch-mean losses IS the gradient of the mean over
    the L samples of the reference's shorter last batch (Demo_RSSS.py:242: DataLoader without drop_last)."""
    from fcd_gan_pytorch_amd.dp import RankStridedBatches
    from fcd_gan_pytorch_amd.steps import _backward
    n, bs = 23, 3
    torch.manual_seed(3)
    data = torch.randn(n, 5)
    w0 = torch.randn(5)
    for world in (2, 4, 8):
        per_rank = [RankStridedBatches(n, bs, seed=4, rank=r, world=world, ragged='weighted') for r in range(world)]
        plans = [list(s) for s in per_rank]
        assert len({len(p) for p in plans}) == 1 and len(plans[0]) == len(per_rank[0])
        last = len(plans[0]) - 1
        assert all(per_rank[r].scales[:last] == [1.0] * last for r in range(world))
        rest = [i for r in range(world) for i in plans[r][last][:len(plans[r][last]) - per_rank[r].pads[last]]]
        L = n - (n // (bs * world)) * bs * world
        assert len(rest) == L and len(set(rest)) == L, 'every remaining tile exactly once, no duplicates'
        assert abs(sum(per_rank[r].scales[last] for r in range(world)) - world) < 1e-12
        for r in range(world):      # a rank left without a tile runs one flagged filler with weight 0
            k = len(plans[r][last]) - per_rank[r].pads[last]
            assert per_rank[r].scales[last] == k * world / float(L) and (k > 0 or (per_rank[r].pads[last] == 1 and len(plans[r][last]) == 1))
        # gradient identity on a batch-mean loss
        wr = w0.clone().requires_grad_(True)
        ((data[rest] @ wr) ** 2).mean().backward()
        acc = torch.zeros(5)
        for r in range(world):
            wl = w0.clone().requires_grad_(True)
            _backward(((data[plans[r][last]] @ wl) ** 2).mean(), per_rank[r].scales[last])
            acc += wl.grad / world                               # what the all-reduce (mean over ranks) leaves
        assert torch.allclose(acc, wr.grad, rtol=1e-5, atol=1e-6)
    one = RankStridedBatches(n, bs, seed=4, rank=0, world=1, ragged='weighted')      # one rank: the reference's loader
    assert [len(b) for b in one] == [3] * 7 + [2] and one.scales == [1.0] * 8 and one.pads == [0] * 8


# ------------------------------------------------- RSSS-shaped exchange through the bucket path
def _rsss_shaped_backward(S, D, optS, optD, x, y, region, literal=False):
    """Control flow of steps.rsss_adversarial_step (minimal mode) on the PRODUCT's parameter objects and
    optimizers, with the oracle's CPU arithmetic standing in for the HIP kernels and the optimizer update (a HIP
    kernel) left out: forward S, D step on the detached map, S step with D frozen."""
    from fcd_gan_pytorch_amd.steps import _frozen
    sdS = dict(S.named_parameters()); sdS.update(dict(S.named_buffers()))
    sdD = dict(D.named_parameters()); sdD.update(dict(D.named_buffers()))
    cmap = onets.segmentor(sdS, x, y, train=False, bilinear=True)
    y_unc = y * (1 - region) + x * region
    keep_d = 1 - cmap.detach()
    c_out = onets.discriminator(sdD, x * keep_d, y * keep_d, train=False)
    nc_out = onets.discriminator(sdD, x * keep_d, y_unc * keep_d, train=False)
    optD.zero_grad()
    d_loss = 1 + nc_out.mean() - c_out.mean()
    armed_d = optD.begin_overlap()
    d_loss.backward()
    optD.allreduce_grads()
    gD, exD = optD.flat_g.clone() * optD.grad_scale, optD.last_exchange
    keep = 1 - cmap
    with _frozen(D):
        c_out = onets.discriminator(sdD, x * keep, y * keep, train=False)
    s_loss = c_out.mean() + 0.02 * (cmap * region).abs().mean() + 2 * ((cmap * (1 - region)) ** 2).mean()
    optS.zero_grad()
    armed_s = optS.begin_overlap()
    s_loss.backward()
    optS.allreduce_grads()
    return gD, optS.flat_g.clone() * optS.grad_scale, exD, optS.last_exchange, (armed_d, armed_s)


def _make_sd(C):
    from fcd_gan_pytorch_amd import Module
    S = Module.Segmentor(C, 1, True); S.load_state_dict(seeded_state(onets.segmentor_spec(C, 1, True), 5)); S.eval()
    D = Module.Discriminator_SRGAN_simple(C); D.load_state_dict(seeded_state(onets.discriminator_spec(C), 6)); D.eval()
    return S, D


def _bucket_worker(rank, world, port, q):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        from fcd_gan_pytorch_amd import optim, dp
        torch.set_num_threads(2)
        C, N = 3, 2
        S, D = _make_sd(C)
        optS, optD = optim.RMSprop(S.parameters(), lr=1e-4), optim.RMSprop(D.parameters(), lr=1e-4)
        if rank == 1:                                   # rank 1 starts from different weights: sync_start must fix that
            with torch.no_grad():
                optS.flat_p.mul_(1.01); optD.flat_p.add_(0.01)
                for b in S.buffers():
                    if b.is_floating_point():
                        b.add_(0.5)
        dp.sync_start((S, D), (optS, optD))
        x, y, region = seeded_tiles(31, N, C, 32, 32)
        sl = slice(rank * N // world, (rank + 1) * N // world)
        gD, gS, exD, exS, armed = _rsss_shaped_backward(S, D, optS, optD, x[sl], y[sl], region[sl])
        q.put((rank, gD.numpy(), gS.numpy(), exD, exS, armed, optS.flat_p.double().sum().item()))
    finally:
        dist.destroy_process_group()


def test_bucketed_overlapped_exchange_rsss_shaped_step():
    """Two gloo ranks x half batch through begin_overlap / gradient-ready hooks / bucketed async all-reduce ==
    the single-process full-batch gradients; buckets are launched while backward is still running; rank 1's
    deliberately perturbed start state is overwritten by rank 0's (dp.sync_start)."""
    world = 2
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_bucket_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=600) for _ in range(world)], key=lambda t: t[0])
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    from fcd_gan_pytorch_amd import optim
    C, N = 3, 2
    x, y, region = seeded_tiles(31, N, C, 32, 32)
    # reference: the same shards run one after the other in THIS process and averaged by hand -- isolates the exchange
    # machinery (oneDNN picks batch-size dependent algorithms, so a literal N=2 run differs from the shard mean by
    # ~1e-3 of the gradient scale through flipped ReLU decisions; checked loosely below)
    shard = []
    for sl in (slice(0, 1), slice(1, 2), slice(0, 2)):
        S, D = _make_sd(C)
        optS, optD = optim.RMSprop(S.parameters(), lr=1e-4), optim.RMSprop(D.parameters(), lr=1e-4)
        gD, gS, exD, exS, armed = _rsss_shaped_backward(S, D, optS, optD, x[sl], y[sl], region[sl])
        assert armed == (False, False) and exS is None                  # single rank: nothing armed, no collective
        shard.append((gD.numpy(), gS.numpy()))
    gD, gS = 0.5 * (shard[0][0] + shard[1][0]), 0.5 * (shard[0][1] + shard[1][1])
    assert np.abs(gS - shard[2][1]).max() <= 5e-3 * np.abs(gS).max() and np.abs(gD - shard[2][0]).max() <= 5e-3 * np.abs(gD).max()
    assert abs(res[0][6] - res[1][6]) == 0.0 and abs(res[0][6] - optS.flat_p.double().sum().item()) < 1e-9
    for rank, rD, rS, eD, eS, ar, _ in res:
        assert ar == (True, True)
        np.testing.assert_allclose(rD, gD, rtol=1e-4, atol=2e-5 * np.abs(gD).max())    # (thread-count dependent summation)
        np.testing.assert_allclose(rS, gS, rtol=1e-4, atol=2e-5 * np.abs(gS).max())
        # Segmentor: 163 MB of gradients in 4 buckets, decoder first; at least the first ones leave during backward
        assert eS['buckets'] == 4 and sum(eS['bytes']) == 4 * optS.flat_g.numel()
        assert eS['bytes'][1] == 4 * (1024 * 2048 * 9 + 3 * 1024)       # up1.conv.double_conv.0.weight (+ bias, BN affine) alone
        assert eS['launched_during_backward'] >= 2, eS
        assert eD['buckets'] == 1
    np.testing.assert_array_equal(res[0][2], res[1][2])                  # both ranks end with the same reduced buffer


# ------------------------------------------------- BatchNorm running statistics before inference / checkpoints
def _buffers_worker(rank, world, port, q):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        from fcd_gan_pytorch_amd import dp, Module
        torch.set_num_threads(1)
        D = Module.Discriminator_SRGAN_simple(3)
        D.load_state_dict(seeded_state(onets.discriminator_spec(3), 6))
        with torch.no_grad():                      # per-replica statistics drift: every rank ends training elsewhere
            for i, b in enumerate(D.buffers()):
                b.add_(rank * (i + 1)) if b.is_floating_point() else b.add_(7 * rank)
        before = [b._version for b in D.buffers()]
        dp.sync_buffers((D,))
        after = [b._version for b in D.buffers()]
        q.put((rank, [b.double().numpy().copy() for b in D.buffers()], all(a > b for a, b in zip(after, before))))
    finally:
        dist.destroy_process_group()


def test_sync_buffers_makes_inference_and_checkpoint_agree():
    """with per-replica BatchNorm the running statistics differ per rank after training; the demos
    broadcast rank 0's before eval-mode inference and before saving, so the stitched map == what the saved
    checkpoint reproduces.  Version counters are bumped (the folded conv+BN caches are keyed by them)."""
    world = 2
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_buffers_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=300) for _ in range(world)], key=lambda t: t[0])
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    from fcd_gan_pytorch_amd import Module
    D = Module.Discriminator_SRGAN_simple(3)
    D.load_state_dict(seeded_state(onets.discriminator_spec(3), 6))
    want = [b.double().numpy() for b in D.buffers()]            # rank 0 added 0
    assert len(want) > 0
    for rank, bufs, bumped in res:
        assert bumped
        for got, w in zip(bufs, want):
            np.testing.assert_array_equal(got, w)


# ------------------------------------------------- world = 8: one epoch per ragged mode through buckets + hooks
def _sample_losses(sd, x, y, idx):
    out = onets.generator(sd, x[idx], train=False)
    return ((out - y[idx]) ** 2).mean()          # eval-mode BN: a plain mean over the local samples


def _world8_worker(rank, world, port, q):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        from fcd_gan_pytorch_amd import optim, dp
        from fcd_gan_pytorch_amd.steps import _backward
        torch.set_num_threads(1)
        C, n, bs = 4, 21, 1
        g = _net_and_state(C)
        opt = optim.Adam(g.parameters(), lr=1e-3, betas=(0.9, 0.99))
        opt.bucket_bytes = 256 << 10             # 1.8 MB of gradients -> several buckets, so the in-order issue logic runs at 8 ranks
        if rank:
            with torch.no_grad():
                opt.flat_p.mul_(1.0 + 0.01 * rank)
        dp.sync_start((g,), (opt,))
        x, y, _ = seeded_tiles(55, n, C, 16, 16)
        sd = dict(g.named_parameters()); sd.update(dict(g.named_buffers()))
        out = {}
        for mode in ('pad', 'drop', 'weighted'):
            sampler = dp.RankStridedBatches(n, bs, seed=7, ragged=mode)          # rank / world from the process group
            grads, early = [], []
            for i, idx in enumerate(sampler):
                opt.zero_grad()
                armed = opt.begin_overlap()
                assert armed
                _backward(_sample_losses(sd, x, y, idx), sampler.scales[i])
                opt.allreduce_grads()
                grads.append((opt.flat_g * opt.grad_scale).clone().numpy())
                early.append(opt.last_exchange['launched_during_backward'])
                nb = opt.last_exchange['buckets']
            out[mode] = (grads, early, nb, [list(b) for b in sampler], list(sampler.pads), list(sampler.scales))
        mean = dp.mean_scalars(torch.tensor([float(rank)]), weight=float(rank + 1))
        q.put((rank, out, float(mean), opt.flat_p.double().sum().item()))
    finally:
        dist.destroy_process_group()


def test_world8_epoch_all_ragged_modes():
    """only world = 2 had ever run.  Eight gloo ranks walk one epoch of 21 tiles (global batch 8: two full
    steps + 5 left over) in each ragged mode through begin_overlap / hooks / several buckets: every rank runs the same number of
    steps (no dead-lock), every step's averaged gradient equals the single-process gradient of the samples that step is
    meant to represent (pad: the padded global batch; drop: full batches only; weighted: exactly the 5 remaining tiles)."""
    world = 8
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_world8_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=900) for _ in range(world)], key=lambda t: t[0])
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    from fcd_gan_pytorch_amd import optim
    C, n = 4, 21
    g = _net_and_state(C)
    opt = optim.Adam(g.parameters(), lr=1e-3, betas=(0.9, 0.99))
    x, y, _ = seeded_tiles(55, n, C, 16, 16)
    sd = dict(g.named_parameters()); sd.update(dict(g.named_buffers()))
    assert len({r[3] for r in res}) == 1 and abs(res[0][3] - opt.flat_p.double().sum().item()) < 1e-9      # sync_start at 8 ranks
    want_mean = sum(r * (r + 1) for r in range(world)) / float(sum(r + 1 for r in range(world)))
    assert all(abs(r[2] - want_mean) < 1e-6 for r in res)

    def full_grad(idx):
        opt.zero_grad()
        _sample_losses(sd, x, y, idx).backward()
        return opt.gather_grads().clone().numpy()
    for mode, nsteps in (('pad', 3), ('drop', 2), ('weighted', 3)):
        per = [r[1][mode] for r in res]
        assert all(len(p[0]) == nsteps for p in per), 'every rank must run the same number of steps (%s)' % mode
        assert all(p[2] >= 4 for p in per)                                          # several buckets
        for s in range(nsteps):
            members = [i for p in per for i in p[3][s]]                              # with duplicates / fillers as dealt
            if mode == 'weighted' and s == nsteps - 1:
                real = [i for p in per for k, i in enumerate(p[3][s]) if p[5][s] > 0]
                assert len(real) == 5 and len(set(real)) == 5 and sum(1 for p in per if p[5][s] == 0.0) == 3
                want = full_grad(real)
            else:
                assert len(members) == world
                want = full_grad(members)
            for rk, p in enumerate(per):
                np.testing.assert_allclose(p[0][s], want, rtol=2e-4, atol=2e-5 * np.abs(want).max(), err_msg='%s step %d rank %d' % (mode, s, rk))
                assert p[1][s] >= 1, 'no bucket left during backward (%s step %d rank %d)' % (mode, s, rk)
            for p in per[1:]:
                np.testing.assert_array_equal(p[0][s], per[0][0][s])                 # identical reduced buffer on all 8 ranks
        if mode == 'pad':
            real = [i for p in per for s in range(nsteps) for i in p[3][s][:len(p[3][s]) - p[4][s]]]
            assert sorted(real) == list(range(n))
