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
        local = opt.flat_g.clone()
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
    full = opt.flat_g.numpy()
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
