"""The plain-C first-principles restatement (oracle/conv_ref.c, fp64 accumulation) agrees with
the ATen operators the oracle is built from -- pins the oracle's own building blocks."""
import ctypes
import os
import subprocess

import numpy as np
import pytest
import torch
import torch.nn.functional as F

ODIR = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'oracle')


@pytest.fixture(scope='module')
def clib():
    subprocess.check_call(['make', '-C', ODIR, '-s'])
    return ctypes.CDLL(os.path.join(ODIR, 'libfcd_oracle_c.so'))


def fp(a):
    return a.ctypes.data_as(ctypes.c_void_p)


@pytest.mark.parametrize('case', [(2, 5, 9, 11, 7, 3, 1, 1), (1, 4, 12, 10, 6, 3, 2, 1), (1, 3, 13, 12, 4, 9, 1, 4),
                                  (2, 6, 5, 5, 3, 1, 1, 0), (1, 4, 8, 6, 5, 2, 2, 0)])
def test_conv_matches_first_principles(clib, case):
    N, C, H, W, K, R, st, pad = case
    rng = np.random.default_rng(sum(case))
    x = rng.standard_normal((N, C, H, W)).astype(np.float32)
    w = (rng.standard_normal((K, C, R, R)) * 0.2).astype(np.float32)
    b = rng.standard_normal(K).astype(np.float32)
    xt, wt = torch.from_numpy(x).requires_grad_(True), torch.from_numpy(w).requires_grad_(True)
    y = F.conv2d(xt, wt, torch.from_numpy(b), stride=st, padding=pad)
    dy = rng.standard_normal(tuple(y.shape)).astype(np.float32)
    y.backward(torch.from_numpy(dy))
    yc = np.zeros(tuple(y.shape), np.float32)
    clib.fcd_ref_conv2d_fwd(fp(x), fp(w), fp(b), fp(yc), N, C, H, W, K, R, R, st, pad)
    dx, dw = np.zeros_like(x), np.zeros_like(w)
    clib.fcd_ref_conv2d_bwd(fp(x), fp(w), fp(dy), fp(dx), fp(dw), N, C, H, W, K, R, R, st, pad)
    np.testing.assert_allclose(y.detach().numpy(), yc, rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(xt.grad.numpy(), dx, rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(wt.grad.numpy(), dw, rtol=1e-4, atol=2e-5)


def test_bn_stats_match_first_principles(clib):
    rng = np.random.default_rng(3)
    x = (rng.standard_normal((3, 5, 6, 7)) * 2 + 0.5).astype(np.float32)
    rm, rv = torch.zeros(5), torch.ones(5)
    F.batch_norm(torch.from_numpy(x), rm, rv, None, None, training=True, momentum=1.0, eps=1e-5)
    mean, var = np.zeros(5), np.zeros(5)
    clib.fcd_ref_bn_stats(fp(x), fp(mean), fp(var), 3, 5, 42)
    n = 3 * 42
    np.testing.assert_allclose(rm.numpy(), mean, rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(rv.numpy(), var * n / (n - 1), rtol=1e-5)      # running_var is unbiased
