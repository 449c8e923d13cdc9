"""Optional SyncBN: two ranks (sharing the one GPU of the test box, gloo transport
for the tiny all-reduces) each holding half the batch reproduce the single-process full-batch
BatchNorm numerics: outputs, input gradients, parameter gradients, running statistics."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket(); s.bind(('127.0.0.1', 0)); p = s.getsockname()[1]; s.close()
    return p


def _make():
    from fcd_gan_pytorch_amd import Module
    torch.manual_seed(3)
    net = Module.DoubleConv(4, 16).cuda().train()
    g = torch.Generator().manual_seed(4)
    x = torch.randn(4, 4, 20, 24, generator=g)
    probe = torch.randn(4, 16, 20, 24, generator=g)
    return net, x, probe


def _worker(rank, world, port, q):
    os.environ['MASTER_ADDR'] = '127.0.0.1'; os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        import fcd_gan_pytorch_amd as p
        p.set_sync_batchnorm(True)
        net, x, probe = _make()
        sl = slice(rank * 2, rank * 2 + 2)
        xg = x[sl].cuda().requires_grad_(True)
        y = net(xg)
        (y * probe[sl].cuda()).sum().backward()
        grads = {k: v.grad.cpu().numpy() for k, v in net.named_parameters()}
        bufs = {k: v.cpu().numpy() for k, v in net.named_buffers()}
        q.put((rank, y.detach().cpu().numpy(), xg.grad.cpu().numpy(), grads, bufs))
    finally:
        dist.destroy_process_group()


def test_syncbn_two_ranks_equal_full_batch():
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for pr in procs:
        pr.start()
    res = sorted([q.get(timeout=600) for _ in range(2)], key=lambda t: t[0])
    for pr in procs:
        pr.join(120)
        assert pr.exitcode == 0
    net, x, probe = _make()
    xg = x.cuda().requires_grad_(True)
    y = net(xg)
    (y * probe.cuda()).sum().backward()
    yf, dxf = y.detach().cpu().numpy(), xg.grad.cpu().numpy()
    for rank, yr, dxr, grads, bufs in res:
        sl = slice(rank * 2, rank * 2 + 2)
        np.testing.assert_allclose(yr, yf[sl], rtol=1e-4, atol=1e-5)
        np.testing.assert_allclose(dxr, dxf[sl], rtol=1e-3, atol=1e-4 * np.abs(dxf).max())
        for k, v in net.named_buffers():
            np.testing.assert_allclose(bufs[k], v.cpu().numpy(), rtol=1e-5, atol=1e-6)
    for k, v in net.named_parameters():
        tot = res[0][3][k] + res[1][3][k]           # local sums add up to the full-batch gradient
        ref = v.grad.cpu().numpy()
        if k.endswith('0.bias') or k.endswith('3.bias'):
            continue                                   # conv bias before BN: analytically zero
        np.testing.assert_allclose(tot, ref, rtol=2e-3, atol=2e-4 * np.abs(ref).max())
