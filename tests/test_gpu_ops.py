"""GPU parity of every C-ABI op against the same ATen op on CPU (the arithmetic the
reference reaches through torch.nn), seeded inputs, fp32.  Tolerances: the MFMA
kernels are exact-fp32 fma chains in a different summation order than oneDNN, so
per-op error is ~1e-6 relative to sum|a*b|; tests use 2e-5 * scale."""
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


def assert_close(got, ref, tol=2e-5, what=''):
    got = got.detach().cpu().double()
    ref = ref.detach().double()
    scale = max(ref.abs().max().item(), 1e-6)
    err = (got - ref).abs().max().item()
    assert got.shape == ref.shape, (what, got.shape, ref.shape)
    assert err <= tol * scale + 1e-7, '%s: max err %.3e (scale %.3e)' % (what, err, scale)


CONV_CASES = [
    # N, C, H, W, K, R, stride, pad
    (2, 4, 32, 32, 64, 3, 1, 1),
    (1, 13, 24, 40, 64, 3, 1, 1),
    (2, 64, 40, 56, 64, 3, 1, 1),
    (1, 64, 16, 16, 128, 3, 1, 1),
    (2, 128, 20, 28, 256, 3, 1, 1),
    (1, 256, 10, 14, 96, 3, 1, 1),
    (3, 24, 5, 7, 40, 3, 1, 1),
    (2, 8, 2, 2, 16, 3, 1, 1),
    (2, 4, 32, 32, 64, 3, 2, 1),
    (3, 64, 24, 20, 128, 3, 2, 1),
    (2, 128, 19, 25, 256, 3, 2, 1),
    (2, 40, 18, 22, 48, 3, 2, 1),       # sub-pixel data gradient: phase blocks of 40 rows straddle the waves' 64-row tiles
    (2, 24, 12, 80, 20, 3, 2, 1),       # ... its LDS-DMA kernel on a wide map (4 x 32 pixel tiles), last chunk of dy channels half full
    (1, 72, 14, 70, 136, 3, 2, 1),      # ... three row tiles, odd output width, 17 chunks
    (2, 4, 32, 32, 64, 9, 1, 4),
    (1, 13, 24, 40, 64, 9, 1, 4),
    (2, 64, 24, 40, 13, 9, 1, 4),
    (2, 64, 20, 36, 4, 9, 1, 4),
    (2, 128, 40, 56, 1, 1, 1, 0),
    (3, 512, 1, 1, 1024, 1, 1, 0),
    (3, 1024, 1, 1, 1, 1, 1, 0),
    (32, 512, 1, 1, 1000, 1, 1, 0),     # small-FC kernel at its sample limit, ragged output block
    (33, 96, 1, 1, 130, 1, 1, 0),       # one sample more: implicit-GEMM tiles
    (2, 3, 48, 40, 64, 3, 1, 1),
    (2, 512, 6, 5, 512, 3, 1, 1),
    # thin (<= 4 input channels) VALU kernels: forward (K > 32) and data gradient, float4 rows
    # (W % 4 == 0, K % 8 == 0) and the generic variants, tile edges in both directions
    (2, 1, 40, 72, 64, 3, 1, 1),
    (2, 1, 35, 128, 64, 3, 1, 1),      # one band, 64 filters, W % 64 == 0: matrix-core data gradient (no mask)
    (1, 1, 9, 256, 64, 3, 1, 1),
    (1, 1, 19, 21, 64, 3, 1, 1),
    (2, 2, 17, 132, 72, 3, 1, 1),
    (1, 4, 33, 68, 40, 3, 1, 1),
    (2, 3, 16, 64, 44, 3, 1, 1),
    (1, 2, 5, 7, 16, 3, 1, 1),
    # first-layer weight gradient (conv_wgrad_thin.hip: C x 9 < 128 columns, N P Q >= 4096): 13 / 4 / 3 bands, stride 1 and 2,
    # rows that are not float4 multiples, ragged tiles in both directions, fewer than 64 filters
    (2, 13, 72, 136, 64, 3, 1, 1),
    (3, 4, 55, 70, 48, 3, 1, 1),
    (2, 13, 96, 130, 64, 3, 2, 1),
    (4, 3, 64, 64, 64, 3, 2, 1),
    (1, 14, 66, 64, 33, 3, 1, 1),
    # 9x9 weight gradient with <= 4 channels on one side (conv_wgrad_thin9.hip: N P Q >= 4096): first-layer form (bands -> filters) and
    # last-layer form (filters -> bands, operands swapped, taps un-flipped), ragged tiles, rows that are not float4 multiples
    (2, 4, 72, 136, 64, 9, 1, 4),
    (3, 3, 55, 70, 48, 9, 1, 4),
    (2, 64, 66, 64, 4, 9, 1, 4),
    (2, 40, 50, 130, 3, 9, 1, 4),
    # filter-resident split GEMM (one 128-row tile, Kc <= 128, >= 512 column-tile x position work items): two and three
    # reduction stages, a ragged last column tile
    (4, 64, 128, 128, 128, 3, 1, 1),
    (6, 96, 100, 116, 128, 3, 1, 1),
]


@pytest.mark.parametrize('case', CONV_CASES, ids=lambda c: 'x'.join(map(str, c)))
def test_conv2d_fwd_bwd(case):
    ops = _ops()
    N, C, H, W, K, R, st, pad = case
    x = rnd(N, C, H, W, seed=1)
    w = rnd(K, C, R, R, seed=2, scale=(2.0 / (C * R * R)) ** 0.5)
    b = rnd(K, seed=3, scale=0.1)
    xr, wr, br = x.clone().requires_grad_(True), w.clone().requires_grad_(True), b.clone().requires_grad_(True)
    yr = F.conv2d(xr, wr, br, stride=st, padding=pad)
    g = rnd(*yr.shape, seed=4)
    yr.backward(g)
    xg, wg, bg = (t.cuda().requires_grad_(True) for t in (x, w, b))
    y = ops.conv2d(xg, wg, bg, st, pad)
    y.backward(g.cuda())
    assert_close(y, yr, what='y')
    assert_close(xg.grad, xr.grad, what='dx')
    assert_close(wg.grad, wr.grad, tol=5e-5, what='dw')
    assert_close(bg.grad, br.grad, what='db')


@pytest.mark.parametrize('case', [(2, 3, 24, 40, 64, 3, 1, 1), (2, 64, 20, 28, 128, 3, 1, 1), (3, 16, 9, 7, 24, 3, 1, 1),
                                  (2, 8, 16, 16, 40, 3, 2, 1), (2, 1, 36, 68, 64, 3, 1, 1), (1, 1, 21, 30, 64, 3, 1, 1),
                                  (2, 13, 40, 72, 64, 3, 1, 1), (2, 4, 90, 102, 64, 3, 2, 1)], ids=lambda c: 'x'.join(map(str, c)))
def test_conv2d_fused_relu(case):
    """conv + bias + ReLU in the kernel epilogue; backward re-derives the mask from the output."""
    ops = _ops()
    N, C, H, W, K, R, st, pad = case
    x = rnd(N, C, H, W, seed=91)
    w = rnd(K, C, R, R, seed=92, scale=(2.0 / (C * R * R)) ** 0.5)
    b = rnd(K, seed=93, scale=0.1)
    xr, wr, br = (t.clone().requires_grad_(True) for t in (x, w, b))
    yr = F.relu(F.conv2d(xr, wr, br, stride=st, padding=pad))
    g = rnd(*yr.shape, seed=94)
    yr.backward(g)
    xg, wg, bg = (t.cuda().requires_grad_(True) for t in (x, w, b))
    y = ops.conv2d(xg, wg, bg, st, pad, relu=True)
    y.backward(g.cuda())
    assert_close(y, yr, what='y')
    # elements whose pre-activation is within rounding of 0 may flip: compare in aggregate
    for got, ref, what in ((xg.grad, xr.grad, 'dx'), (wg.grad, wr.grad, 'dw'), (bg.grad, br.grad, 'db')):
        d = (got.cpu().double() - ref.double())
        assert (d.norm() / ref.double().norm()).item() < 1e-4, what


@pytest.mark.parametrize('case', [(2, 64, 40, 64, 64), (1, 64, 24, 40, 128), (2, 128, 16, 16, 256), (1, 64, 13, 27, 64),
                                  (2, 40, 8, 8, 48), (1, 64, 33, 70, 96)], ids=lambda c: 'x'.join(map(str, c)))
def test_conv_relu_maxpool_fused(case):
    """conv3x3+bias+ReLU+MaxPool2d(2) in one kernel vs the ATen composition, forward and data
    gradient.  The input is built so that the branchy cases really occur: exact ties inside pooling
    windows (first maximum must win), windows that are entirely <= 0, odd map sizes."""
    ops = _ops()
    N, C, H, W, K = case
    x = rnd(N, C, H, W, seed=101)
    x[:, :, : H // 2, : W // 2] = x[:, :, :1, :1]          # constant block => many equal conv outputs (ties)
    w = rnd(K, C, 3, 3, seed=102, scale=(2.0 / (C * 9)) ** 0.5)
    b = rnd(K, seed=103, scale=0.1)
    b[: K // 4] = -50.0                                     # a quarter of the channels: all windows negative
    xr = x.clone().requires_grad_(True)
    yr = F.max_pool2d(F.relu(F.conv2d(xr, w, b, padding=1)), 2)
    g = rnd(*yr.shape, seed=104)
    yr.backward(g)
    xg = x.cuda().requires_grad_(True)
    wg, bg = w.cuda(), b.cuda()
    assert ops.conv_relu_pool_supported(xg, wg)
    y = ops.conv2d_relu_maxpool2(xg, wg, bg)
    y.backward(g.cuda())
    assert_close(y, yr, what='pooled y')
    d = xg.grad.cpu().double() - xr.grad.double()
    # near-ties (|difference| at rounding level) may pick the other slot: aggregate bound + exact bulk
    assert (d.norm() / xr.grad.double().norm().clamp_min(1e-30)).item() < 2e-3
    assert (d.abs() > 1e-4 * xr.grad.abs().max().item()).double().mean().item() < 5e-3


@pytest.mark.parametrize('shape', [(2, 16, 5, 7), (1, 64, 12, 16), (3, 32, 1, 1)])
def test_conv_transpose2x2(shape):
    ops = _ops()
    N, Cin, h, w = shape
    Cout = Cin // 2
    x, wt, b = rnd(N, Cin, h, w, seed=5), rnd(Cin, Cout, 2, 2, seed=6, scale=0.2), rnd(Cout, seed=7, scale=0.1)
    xr, wr, br = (t.clone().requires_grad_(True) for t in (x, wt, b))
    yr = F.conv_transpose2d(xr, wr, br, stride=2)
    g = rnd(*yr.shape, seed=8)
    yr.backward(g)
    xg, wg, bg = (t.cuda().requires_grad_(True) for t in (x, wt, b))
    y = ops.conv_transpose2x2(xg, wg, bg)
    y.backward(g.cuda())
    assert_close(y, yr, what='y')
    assert_close(xg.grad, xr.grad, what='dx')
    assert_close(wg.grad, wr.grad, tol=5e-5, what='dw')
    assert_close(bg.grad, br.grad, what='db')


class _BN:
    pass


def _mk_bn(C, training, seed):
    bn = torch.nn.BatchNorm2d(C)
    with torch.no_grad():
        bn.weight.copy_(1 + 0.1 * rnd(C, seed=seed))
        bn.bias.copy_(0.1 * rnd(C, seed=seed + 1))
        bn.running_mean.copy_(0.1 * rnd(C, seed=seed + 2))
        bn.running_var.copy_(1 + 0.2 * torch.tanh(rnd(C, seed=seed + 3)))
    bn.train(training)
    return bn


@pytest.mark.parametrize('training', [True, False])
@pytest.mark.parametrize('act', ['none', 'relu', 'leaky', 'prelu'])
@pytest.mark.parametrize('shape,groups', [((4, 16, 12, 20), 1), ((4, 8, 9, 7), 2), ((8, 32, 4, 4), 4), ((2, 64, 33, 1), 1)])
def test_bn_act(training, act, shape, groups):
    ops = _ops()
    import copy
    N, C, H, W = shape
    x = rnd(*shape, seed=11) * 1.5 + 0.3
    g = rnd(*shape, seed=12)
    bn_ref = _mk_bn(C, training, 20)
    bn_gpu = copy.deepcopy(bn_ref).cuda()
    slope_ref = torch.tensor([0.25], requires_grad=True)
    slope_gpu = torch.tensor([0.25], device='cuda', requires_grad=True)

    def actf(t, sl):
        if act == 'relu':
            return F.relu(t)
        if act == 'leaky':
            return F.leaky_relu(t, 0.2)
        if act == 'prelu':
            return F.prelu(t, sl)
        return t
    xr = x.clone().requires_grad_(True)
    Ng = N // groups
    outs = [actf(bn_ref(xr[i * Ng:(i + 1) * Ng]), slope_ref) for i in range(groups)]   # sequential calls
    yr = torch.cat(outs, 0)
    yr.backward(g)
    code = {'none': ops.ACT_NONE, 'relu': ops.ACT_RELU, 'leaky': ops.ACT_LEAKY, 'prelu': ops.ACT_PRELU}[act]
    xg = x.cuda().requires_grad_(True)
    y = ops.bn_act(xg, bn_gpu, code, slope=slope_gpu if act == 'prelu' else None, slope_imm=0.2, groups=groups)
    y.backward(g.cuda())
    assert_close(y, yr, what='y')
    assert_close(xg.grad, xr.grad, tol=5e-5, what='dx')
    assert_close(bn_gpu.weight.grad, bn_ref.weight.grad, tol=5e-5, what='dgamma')
    assert_close(bn_gpu.bias.grad, bn_ref.bias.grad, tol=5e-5, what='dbeta')
    if act == 'prelu':
        assert_close(slope_gpu.grad, slope_ref.grad, tol=5e-5, what='dslope')
    assert_close(bn_gpu.running_mean, bn_ref.running_mean, what='running_mean')
    assert_close(bn_gpu.running_var, bn_ref.running_var, what='running_var')
    assert int(bn_gpu.num_batches_tracked) == int(bn_ref.num_batches_tracked)


@pytest.mark.parametrize('act', ['leaky', 'prelu', 'relu'])
def test_bare_activation(act):
    ops = _ops()
    x, g = rnd(2, 8, 7, 9, seed=31), rnd(2, 8, 7, 9, seed=32)
    sr = torch.tensor([0.3], requires_grad=True)
    sg = torch.tensor([0.3], device='cuda', requires_grad=True)
    xr = x.clone().requires_grad_(True)
    yr = {'leaky': lambda t: F.leaky_relu(t, 0.2), 'prelu': lambda t: F.prelu(t, sr), 'relu': F.relu}[act](xr)
    yr.backward(g)
    xg = x.cuda().requires_grad_(True)
    code = {'relu': ops.ACT_RELU, 'leaky': ops.ACT_LEAKY, 'prelu': ops.ACT_PRELU}[act]
    y = ops.bn_act(xg, None, code, slope=sg if act == 'prelu' else None, slope_imm=0.2)
    y.backward(g.cuda())
    assert_close(y, yr, what='y')
    assert_close(xg.grad, xr.grad, what='dx')
    if act == 'prelu':
        assert_close(sg.grad, sr.grad, tol=5e-5, what='dslope')


@pytest.mark.parametrize('case', [(2, 128, 64, 64, True), (3, 21, 32, 36, True), (1, 8, 32, 32, False), (2, 130, 48, 40, True)],
                         ids=lambda c: 'x'.join(map(str, c)))
def test_conv1x1_head(case):
    """One-output-channel 1x1 convolution + sigmoid (csrc/conv_head.hip; reference Module.py:82-90 OutConv) vs the ATen
    composition in fp64: forward, dx, dw, db; channel counts off the unroll of 8, with and without the sigmoid, and the
    frozen-filter case (dx only)."""
    ops = _ops()
    if not ops._lib.switch('CONV_HEAD'):
        pytest.skip('head kernels switched off')
    N, C, H, W, sig = case
    x, w, b, g = rnd(N, C, H, W, seed=71), rnd(1, C, 1, 1, seed=72, scale=C ** -0.5), rnd(1, seed=73), rnd(N, 1, H, W, seed=74)
    xr, wr, br = (t.double().requires_grad_(True) for t in (x, w, b))
    yr = F.conv2d(xr, wr, br)
    yr = torch.sigmoid(yr) if sig else yr
    yr.backward(g.double())
    xg, wg, bg = (t.cuda().requires_grad_(True) for t in (x, w, b))
    assert ops.conv1x1_head_supported(xg, wg)
    y = ops.conv1x1_head(xg, wg, bg, sigmoid=sig)
    y.backward(g.cuda())
    assert_close(y, yr, what='y')
    assert_close(xg.grad, xr.grad, what='dx')
    assert_close(wg.grad, wr.grad, what='dw')
    assert_close(bg.grad, br.grad, what='db')
    xf = x.cuda().requires_grad_(True)                     # frozen filter: no weight-gradient pass, x is not kept
    ops.conv1x1_head(xf, w.cuda(), b.cuda(), sigmoid=sig).backward(g.cuda())
    assert_close(xf.grad, xr.grad, what='dx (frozen filter)')
    assert not ops.conv1x1_head_supported(rnd(2, 16, 1, 1).cuda(), rnd(1, 16, 1, 1).cuda())       # 1 x 1 maps: small-FC kernel
    assert not ops.conv1x1_head_supported(xg, rnd(2, C, 1, 1).cuda())                              # two output channels


@pytest.mark.parametrize('shape', [(2, 3, 8, 8), (1, 5, 11, 13), (2, 4, 27, 55), (1, 2, 2, 2), (1, 2, 3, 3), (1, 3, 5, 4), (2, 2, 9, 12),
                                   (1, 2, 32, 64)])
def test_maxpool_upsample_avgpool(shape):
    ops = _ops()
    x = rnd(*shape, seed=41)
    for name, ref_fn, fn in (
            ('maxpool', lambda t: F.max_pool2d(t, 2), ops.maxpool2),
            ('upsample', lambda t: F.interpolate(t, scale_factor=2, mode='bilinear', align_corners=True), ops.upsample2x),
            ('avgpool', lambda t: F.avg_pool2d(t, kernel_size=2, padding=[s % 2 for s in t.shape[2:]]), ops.avgpool2_pad)):
        xr = x.clone().requires_grad_(True)
        yr = ref_fn(xr)
        g = rnd(*yr.shape, seed=42)
        yr.backward(g)
        xg = x.cuda().requires_grad_(True)
        y = fn(xg)
        y.backward(g.cuda())
        assert_close(y, yr, tol=1e-6, what=name + ' y')
        assert_close(xg.grad, xr.grad, tol=2e-6, what=name + ' dx')


@pytest.mark.parametrize('shape', [(3, 2, 4), (2, 7, 12), (4, 16, 16), (1, 33, 64), (5, 20, 24), (2, 64, 128), (5, 36, 40), (3, 32, 8)], ids=lambda c: 'x'.join(map(str, c)))
def test_upsample_bwd_four_column_kernel_is_bit_identical(shape):
    """The adjoint of the x2 bilinear upsampling: the four-columns-per-thread kernels (W % 4 == 0, 16-B aligned buffers; from H = 32 up
    the one whose threads walk rows with their column weights kept, row runs crossing plane boundaries in the cases) against the
    one-pixel kernel the same entry point falls back to for a misaligned gradient buffer -- same weights, same summation order."""
    import ctypes
    from fcd_gan_pytorch_amd._lib import lib, check
    ops = _ops()
    NC, H, W = shape
    g = rnd(NC, 2 * H, 2 * W, seed=43).cuda()
    gm = torch.empty(g.numel() + 1, device='cuda')[1:]          # 4-B aligned only
    gm.copy_(g.reshape(-1))
    a, b = torch.empty(NC, H, W, device='cuda'), torch.empty(NC, H, W, device='cuda')
    check(lib.fcd_upsample2x_bwd(ops._p(g), ops._p(a), NC, H, W, ops._stream()))
    check(lib.fcd_upsample2x_bwd(ctypes.c_void_p(gm.data_ptr()), ops._p(b), NC, H, W, ops._stream()))
    torch.cuda.synchronize()
    assert torch.equal(a, b)


@pytest.mark.parametrize('kind', [0, 1])
@pytest.mark.parametrize('complement', [True, False])
def test_masked_sums(kind, complement):
    ops = _ops()
    N, C, H, W = 3, 5, 17, 23
    a, b = rnd(N, C, H, W, seed=51), rnd(N, C, H, W, seed=52)
    m = torch.sigmoid(rnd(N, 1, H, W, seed=53))
    ar, br_, mr = (t.clone().requires_grad_(True) for t in (a, b, m))
    w = (1 - mr) if complement else mr
    d = (ar - br_) * w
    num = (d.abs() if kind == 0 else d * d).sum((1, 2, 3))
    ws = w.sum((1, 2, 3))
    coef, cw = rnd(N, seed=54), rnd(N, seed=55)
    (num * coef + ws * cw).sum().backward()
    ag, bg, mg = (t.cuda().requires_grad_(True) for t in (a, b, m))
    n2, w2 = ops.masked_sums(ag, bg, mg, kind, complement)
    (n2 * coef.cuda() + w2 * cw.cuda()).sum().backward()
    assert_close(n2, num, what='num')
    assert_close(w2, ws, what='wsum')
    assert_close(ag.grad, ar.grad, what='da')
    assert_close(bg.grad, br_.grad, what='db')
    assert_close(mg.grad, mr.grad, what='dm')


@pytest.mark.parametrize('kind', [0, 1])
@pytest.mark.parametrize('skip_zero', [True, False])
def test_masked_ratio_mean(kind, skip_zero):
    """mean_i(num_i * scale / wsum_i) of the masked per-sample sums (reference Loss.py:82-84,115-119,135-138) as one autograd
    node, against the same expression through fp64 autograd; with skip_zero one sample has an all-zero weight map (the
    reference's `continue`) and still counts in the mean."""
    ops = _ops()
    N, C, H, W = 4, 5, 17, 23
    a, b = rnd(N, C, H, W, seed=54), rnd(N, C, H, W, seed=55)
    m = torch.sigmoid(rnd(N, 1, H, W, seed=56))
    if skip_zero:
        m[2] = 0.0                                     # w = m (complement False): sample 2 has wsum == 0
    scale = 1.0 / C
    ar, br_, mr = (t.double().requires_grad_(True) for t in (a, b, m))
    d = (ar - br_) * mr
    num = (d.abs() if kind == 0 else d * d).sum((1, 2, 3))
    ws = mr.sum((1, 2, 3))
    if skip_zero:
        ok = ws != 0
        terms = torch.where(ok, num * scale / torch.where(ok, ws, torch.ones_like(ws)), torch.zeros_like(num))
    else:
        terms = num * scale / ws
    ref = terms.sum() / N
    (ref * 1.7).backward()
    ag, bg, mg = (t.cuda().requires_grad_(True) for t in (a, b, m))
    out = ops.masked_ratio_mean(ag, bg, mg, kind, False, scale, skip_zero)
    (out * 1.7).backward()
    assert out.shape == ()
    assert_close(out, ref, tol=2e-6, what='loss')
    assert_close(ag.grad, ar.grad, tol=2e-6, what='da')
    assert_close(bg.grad, br_.grad, tol=2e-6, what='db')
    assert_close(mg.grad, mr.grad, tol=2e-6, what='dm')
    # one-tensor form (region_loss: b = None, complement weights)
    cg = a[:, :1].contiguous().cuda().requires_grad_(True)
    o2 = ops.masked_ratio_mean(cg, None, m.cuda(), kind, True, 1.0, True)
    cr = a[:, :1].double().requires_grad_(True)
    w2 = 1 - m.double()
    d2 = cr * w2
    r2 = ((d2.abs() if kind == 0 else d2 * d2).sum((1, 2, 3)) / w2.sum((1, 2, 3))).mean()
    o2.backward(); r2.backward()
    assert_close(o2, r2, tol=2e-6, what='loss (one tensor)')
    assert_close(cg.grad, cr.grad, tol=2e-6, what='dcmap')


def test_adam_rmsprop_match_torch():
    ops = _ops()
    n = 10007
    p0, gs = rnd(n, seed=61), [rnd(n, seed=62 + i) * (0.1 if i else 1.0) for i in range(3)]
    for kind in ('adam', 'rmsprop'):
        pr = p0.clone().requires_grad_(True)
        opt = torch.optim.Adam([pr], lr=2e-4, betas=(0.9, 0.99)) if kind == 'adam' else torch.optim.RMSprop([pr], lr=5e-5)
        pg = p0.cuda()
        s1, s2 = torch.zeros_like(pg), torch.zeros_like(pg)
        for step, g in enumerate(gs, 1):
            pr.grad = g.clone()
            opt.step()
            if kind == 'adam':
                ops.adam_step(pg, g.cuda(), s1, s2, 2e-4, 0.9, 0.99, 1e-8, 0.0, step)
            else:
                ops.rmsprop_step(pg, g.cuda(), s1, 5e-5, 0.99, 1e-8, 0.0)
        assert_close(pg, pr, tol=1e-6, what=kind)


@pytest.mark.parametrize('shape', [(2, 3, 40, 52), (1, 4, 176, 176), (1, 2, 200, 184), (2, 1, 11, 11), (1, 1, 13, 31),
                                   (2, 3, 8, 40), (1, 2, 37, 6), (1, 1, 5, 7)])      # [r4] a dimension shorter than the 11-tap window
def test_ssim_level_and_msssim(shape):
    """fused SSIM level vs the oracle's op-by-op restatement of ssim.py:55-92 (incl. gaussian_filter's rule of skipping the
    smoothing along a dimension shorter than the window, ssim.py:44-50)."""
    ops = _ops()
    from oracle import losses as ol
    from fcd_gan_pytorch_amd import ssim as pssim
    N, C, H, W = shape
    x = rnd(*shape, seed=71)
    y = x + 0.3 * rnd(*shape, seed=72)
    g = ol.gauss_window()
    xr, yr = x.clone().requires_grad_(True), y.clone().requires_grad_(True)
    s_ref, cs_ref = ol.ssim_level(xr, yr, g)
    ws, wc = rnd(N, C, seed=73), rnd(N, C, seed=74)
    ((s_ref * ws).sum() + (cs_ref * wc).sum()).backward()
    xg, yg = x.cuda().requires_grad_(True), y.cuda().requires_grad_(True)
    s, cs = ops.ssim_level(xg, yg, g.cuda(), 0.01 ** 2, 0.03 ** 2)
    ((s * ws.cuda()).sum() + (cs * wc.cuda()).sum()).backward()
    assert_close(s, s_ref, tol=2e-5, what='ssim')
    assert_close(cs, cs_ref, tol=2e-5, what='cs')
    assert_close(xg.grad, xr.grad, tol=1e-4, what='dX')
    assert_close(yg.grad, yr.grad, tol=1e-4, what='dY')
    if min(H, W) < 11:       # the public entry point warns like the reference does and returns the same number
        import warnings as _w
        with _w.catch_warnings(record=True) as rec:
            _w.simplefilter('always')
            v = pssim.ssim(x.cuda(), y.cuda(), data_range=1.0)
        assert any('Skipping Gaussian Smoothing' in str(r.message) for r in rec)
        assert abs(v.item() - s_ref.mean().item()) < 2e-5
    if min(H, W) > 160:
        xr.grad = None; yr.grad = None; xg.grad = None; yg.grad = None
        v_ref = ol.ms_ssim(xr, yr, data_range=1.0)
        v_ref.backward()
        v = pssim.MS_SSIM(data_range=1.0, channel=C)(xg, yg)
        v.backward()
        assert abs(v.item() - v_ref.item()) < 1e-5
        assert_close(xg.grad, xr.grad, tol=1e-4, what='ms dX')
        assert_close(yg.grad, yr.grad, tol=1e-4, what='ms dY')


WINO_CASES = [
    # N, C, H, W, K   (3x3 / stride 1 / pad 1, >= 256 reduction channels in at least one direction)
    (2, 256, 16, 16, 256),
    (1, 512, 9, 14, 128),      # P, Q not multiples of the tile
    (3, 256, 33, 21, 384),     # odd sizes, ragged GEMM row / column tiles
    (2, 320, 8, 12, 160),      # data gradient on the direct kernel (160 reduction channels), forward Winograd
    (1, 256, 5, 70, 256),      # more than one 64-column strip per tile row
]


@pytest.mark.parametrize('m', [2, 4])
@pytest.mark.parametrize('case', WINO_CASES, ids=lambda c: 'x'.join(map(str, c)))
def test_conv_winograd_path(case, m):
    """Winograd F(m x m, 3 x 3) forward / data gradient (plain, ReLU-gated, pooled-gradient sources;
    bias / ReLU / max-pool epilogues) vs torch fp64.  fp32 transform rounding: ~3e-7 (m = 2), ~1e-5 (m = 4)."""
    import ctypes
    ops = _ops()
    lib = ops.lib
    N, C, H, W, K = case
    tol = 3e-6 if m == 2 else 6e-5
    flip_tol = 4e-3 if m == 2 else 2e-2     # ReLU / argmax decisions within the forward rounding may fall the other way
    prev = lib.fcd_conv_wino_set(m)
    try:
        d = ops._desc((N, C, H, W), (K, C, 3, 3), 1, 1)
        assert lib.fcd_conv_wino_plan(ctypes.byref(d), 0) == m
        x = rnd(N, C, H, W, seed=11)
        w = rnd(K, C, 3, 3, seed=12, scale=(2.0 / (C * 9)) ** 0.5)
        b = rnd(K, seed=13, scale=0.1)
        g = rnd(N, K, H, W, seed=14)
        xr, wr, br = (t.double().requires_grad_(True) for t in (x, w, b))
        yr = F.conv2d(xr, wr, br, padding=1)
        yr.backward(g.double())
        xg, wg, bg = (t.cuda().requires_grad_(True) for t in (x, w, b))
        y = ops.conv2d(xg, wg, bg, 1, 1)
        y.backward(g.cuda())
        assert_close(y, yr.float(), tol=tol, what='y')
        assert_close(xg.grad, xr.grad.float(), tol=tol, what='dx')
        assert_close(wg.grad, wr.grad.float(), tol=5e-5, what='dw')
        # bit-reproducible
        y2 = ops.conv2d(xg, wg, bg, 1, 1)
        assert torch.equal(y2, y)
        # fused ReLU (+ mask-gated data gradient) and fused ReLU + max-pool (+ code-routed data gradient)
        xr2 = x.double().requires_grad_(True)
        ypr = F.max_pool2d(F.relu(F.conv2d(xr2, w.double(), b.double(), padding=1)), 2)
        gp = rnd(*ypr.shape, seed=15)
        ypr.backward(gp.double())
        xg2 = x.cuda().requires_grad_(True)
        yp = ops.conv2d_relu_maxpool2(xg2, w.cuda(), b.cuda())
        yp.backward(gp.cuda())
        assert_close(yp, ypr.float(), tol=tol, what='pooled y')
        dd = xg2.grad.cpu().double() - xr2.grad
        assert (dd.norm() / xr2.grad.norm()).item() < flip_tol, 'pooled dx'
        xr3 = x.double().requires_grad_(True)
        F.relu(F.conv2d(xr3, w.double(), b.double(), padding=1)).backward(g.double())
        xg3 = x.cuda().requires_grad_(True)
        ops.conv2d(xg3, w.cuda(), b.cuda(), 1, 1, relu=True).backward(g.cuda())
        dd = xg3.grad.cpu().double() - xr3.grad
        assert (dd.norm() / xr3.grad.norm()).item() < flip_tol, 'relu dx'
    finally:
        lib.fcd_conv_wino_set(prev)


@pytest.mark.parametrize('case', [(2, 256, 32, 32, 256), (1, 512, 36, 20, 384), (3, 128, 24, 40, 320), (1, 1024, 16, 16, 512)],
                         ids=lambda c: 'x'.join(map(str, c)))
def test_wino_gemm_matrix_pipes(case):
    """The batched GEMM of the F(4x4,3x3) path on the fp32 matrix pipe (v_mfma_f32_32x32x2_f32) and on the bf16 pipe with
    exact three-way operand splitting (128 x 128 and 256 x 256 tile kernels): forward and data gradient vs torch fp64.
    The split path is fp32-equivalent: same bound as the fp32 pipe, and its error may not exceed the fp32 pipe's by
    more than a rounding's worth."""
    ops = _ops()
    lib = ops.lib
    N, C, H, W, K = case
    x = rnd(N, C, H, W, seed=31)
    w = rnd(K, C, 3, 3, seed=32, scale=(2.0 / (C * 9)) ** 0.5)
    b = rnd(K, seed=33, scale=0.1)
    g = rnd(N, K, H, W, seed=34)
    xr = x.double().requires_grad_(True)
    yr = F.conv2d(xr, w.double(), b.double(), padding=1)
    yr.backward(g.double())
    ys, ds = yr.abs().max().item(), xr.grad.abs().max().item()
    err = {}
    prev = lib.fcd_conv_wino_split_set(-1)
    try:
        for mode in (0, 1, 2):
            lib.fcd_conv_wino_split_set(mode)
            xg = x.cuda().requires_grad_(True)
            y = ops.conv2d(xg, w.cuda(), b.cuda(), 1, 1)
            y.backward(g.cuda())
            ey = (y.detach().cpu().double() - yr.detach())
            ed = (xg.grad.cpu().double() - xr.grad)
            err[mode] = (ey.abs().max().item() / ys, ey.pow(2).mean().sqrt().item() / ys,
                         ed.abs().max().item() / ds, ed.pow(2).mean().sqrt().item() / ds)
            assert err[mode][0] < 6e-5 and err[mode][2] < 6e-5, (mode, err[mode])
    finally:
        lib.fcd_conv_wino_split_set(prev)
    for mode in (1, 2):
        for q in (1, 3):      # rms error of y and of dx (the F(4x4) transforms' own rounding dominates both)
            assert err[mode][q] <= 1.15 * err[0][q] + 1e-8, (mode, q, err)


@pytest.mark.parametrize('case', [(2, 128, 32, 32, 256), (1, 256, 20, 36, 128), (3, 128, 13, 18, 128), (2, 512, 16, 16, 512)],
                         ids=lambda c: 'x'.join(map(str, c)))
def test_wino_frozen_relu_bit_mask(case, switches):
    """Frozen F(4x4) layer with fused ReLU (the VGG16 stack of the perception term): the backward mask kept as 16 sign bits
    per output tile (fcd_conv2d_fwd_wino_relu_bits / fcd_conv2d_bwd_data_wino_bits) must give the SAME data gradient, bit
    for bit, as gating with the fp32 activation -- on all three gated input-transform kernels (strip / 2 x 8 / 4 x 4 blocks),
    ragged maps included -- and both must match torch."""
    ops = _ops()
    N, C, H, W, K = case
    x = rnd(N, C, H, W, seed=81)
    w = rnd(K, C, 3, 3, seed=82, scale=(2.0 / (C * 9)) ** 0.5)
    b = rnd(K, seed=83, scale=0.1)
    g = rnd(N, K, H, W, seed=84)
    d = ops._desc(x.shape, w.shape, 1, 1)
    import ctypes
    if not ops.lib.fcd_conv_wino_relu_bits_bytes(ctypes.byref(d)):
        pytest.skip('layer is not planned F(4x4) in both directions')
    res = {}
    for tag, env in (('bits', '1'), ('y', '0')):
        switches('WINO_RELU_BITS', int(env))
        xg = x.cuda().requires_grad_(True)
        wg = w.cuda()                                   # frozen
        y = ops.conv2d(xg, wg, b.cuda(), 1, 1, relu=True)
        y.backward(g.cuda())
        res[tag] = (y.detach().cpu(), xg.grad.cpu())
    assert torch.equal(res['bits'][0], res['y'][0]) and torch.equal(res['bits'][1], res['y'][1])
    xr = x.double().requires_grad_(True)
    yr = F.relu(F.conv2d(xr, w.double(), b.double(), padding=1))
    yr.backward(g.double())
    assert_close(res['bits'][0], yr, tol=6e-5, what='y')
    dd = res['bits'][1].double() - xr.grad
    assert (dd.norm() / xr.grad.norm()).item() < 2e-4           # (outputs within rounding of 0 may gate the other way)


@pytest.mark.parametrize('case', [(4, 64, 64, 64, 128, 2), (2, 128, 128, 128, 128, 1), (4, 128, 32, 64, 256, 2)],
                         ids=lambda c: 'x'.join(map(str, c)))
def test_bn_statistics_from_the_output_transform(case):
    """Conv2d -> BatchNorm2d(train) -> ReLU (reference Module.py:25-31): the BatchNorm's statistics summed by the F(4x4) output
    transform (conv2d(bn_groups=...) + fcd_bn_act_fwd_parts) against the separate statistics pass over y -- outputs, saved
    statistics through the backward pass (input / filter / affine gradients) and the running statistics, Siamese sample
    groups included.  Both sum in fp64; they differ only in summation order."""
    ops = _ops()
    N, C, H, W, K, G = case
    import ctypes
    d = ops._desc((N, C, H, W), (K, C, 3, 3), 1, 1)
    if not ops.lib.fcd_conv_wino_bn_split(ctypes.byref(d), G):
        pytest.skip('statistics not available from this layer / grouping')
    x = rnd(N, C, H, W, seed=91)
    w = rnd(K, C, 3, 3, seed=92, scale=(2.0 / (C * 9)) ** 0.5)
    b = rnd(K, seed=93, scale=0.1)
    g = rnd(N, K, H, W, seed=94)
    res = {}
    for tag, env in (('fused', '1'), ('separate', '0')):
        bn = torch.nn.BatchNorm2d(K).cuda().train()
        with torch.no_grad():
            bn.weight.copy_(rnd(K, seed=95).cuda() * 0.2 + 1.0)
            bn.bias.copy_(rnd(K, seed=96).cuda() * 0.1)
        xg, wg, bg = (t.cuda().requires_grad_(True) for t in (x, w, b))
        y = ops.conv2d(xg, wg, bg, 1, 1, bn_groups=G if env == '1' else 0)
        assert (getattr(y, '_fcd_bn', None) is not None) == (env == '1')
        z = ops.bn_act(y, bn, ops.ACT_RELU, groups=G)
        z.backward(g.cuda())
        res[tag] = [t.detach().cpu().double() for t in (z, xg.grad, wg.grad, bn.weight.grad, bn.bias.grad, bn.running_mean, bn.running_var)]
        assert int(bn.num_batches_tracked) == G
    for a, r, what in zip(res['fused'], res['separate'], ('z', 'dx', 'dw', 'dgamma', 'dbeta', 'running_mean', 'running_var')):
        err = (a - r).abs().max().item() / max(r.abs().max().item(), 1e-12)
        assert err < 2e-6, (what, err)


@pytest.mark.parametrize('tagged', [False, True])
@pytest.mark.parametrize('case', [(4, 128, 32, 32, 128, 2), (2, 128, 64, 64, 256, 1), (2, 96, 40, 36, 128, 2), (3, 256, 24, 72, 128, 1),
                                  (2, 128, 128, 128, 128, 1), (4, 64, 64, 64, 128, 2)], ids=lambda c: 'x'.join(map(str, c)))
def test_bn_relu_applied_by_the_next_convolutions_loader(case, tagged):
    """Conv2d -> BatchNorm2d(train) -> ReLU -> Conv2d (reference Module.py:25-31, the middle of DoubleConv): ``ops.bn_relu_conv3x3``
    -- the BatchNorm + ReLU pass done by the F(4x4) input transform of the second convolution, the activation never written --
    against ``conv2d(bn_act(z))``.  The loader computes what the apply kernel computes (fma, compare), padding stays zero, the
    backward pass runs the same kernels on the same operands: outputs, all gradients (z, gamma, beta, filter, bias), running
    statistics and the partial sums for the BatchNorm behind must be BIT-identical.  ``tagged``: z comes out of a convolution
    whose output transform summed the statistics (Siamese sample groups included); ragged channel chunk and odd tile counts in
    the cases."""
    ops = _ops()
    N, C, H, W, K, G = case
    zsrc = rnd(N, C, H, W, seed=101)
    w0 = rnd(C, C, 3, 3, seed=100, scale=(2.0 / (C * 9)) ** 0.5)
    w = rnd(K, C, 3, 3, seed=102, scale=(2.0 / (C * 9)) ** 0.5)
    b = rnd(K, seed=103, scale=0.1)
    g = rnd(N, K, H, W, seed=104)
    res = {}
    for tag in ('fused', 'separate'):
        bn = torch.nn.BatchNorm2d(C).cuda().train()
        with torch.no_grad():
            bn.weight.copy_(rnd(C, seed=105).cuda() * 0.2 + 1.0)
            bn.bias.copy_(rnd(C, seed=106).cuda() * 0.1)
        zin = zsrc.cuda().requires_grad_(True)
        wg, bg = (t.cuda().requires_grad_(True) for t in (w, b))
        if tagged:
            z = ops.conv2d(zin, w0.cuda(), None, 1, 1, bn_groups=G)
            if getattr(z, '_fcd_bn', None) is None:
                pytest.skip('no statistics from the producing layer at this shape')
        else:
            z = zin * 1.0
        if tag == 'fused':
            if not ops.bn_relu_conv3x3_ok(z, bn, wg, G):
                pytest.skip('layer does not take the fused loader')
            y = ops.bn_relu_conv3x3(z, bn, wg, bg, groups=G, bn_groups=G)
        else:
            y = ops.conv2d(ops.bn_act(z, bn, ops.ACT_RELU, groups=G), wg, bg, 1, 1, bn_groups=G)
        part = getattr(y, '_fcd_bn', None)
        y.backward(g.cuda())
        res[tag] = [t.detach().cpu() for t in (y, zin.grad, wg.grad, bg.grad, bn.weight.grad, bn.bias.grad, bn.running_mean,
                                               bn.running_var)] + ([part[0].view(-1, 3)[:, :2].cpu()] if part is not None else [])   # {sum, sum of squares, unused}
        assert int(bn.num_batches_tracked) == G
    assert len(res['fused']) == len(res['separate'])
    for a, r, what in zip(res['fused'], res['separate'], ('y', 'dz', 'dw', 'db', 'dgamma', 'dbeta', 'running_mean', 'running_var', 'bn_part')):
        assert torch.equal(a, r), (what, (a.double() - r.double()).abs().max().item())


@pytest.mark.parametrize('shape', [(2, 3, 8, 8), (1, 5, 11, 13), (2, 4, 27, 55), (3, 16, 64, 64), (1, 2, 2, 2)], ids=lambda c: 'x'.join(map(str, c)))
@pytest.mark.parametrize('use', ['both', 'skip_only', 'pool_only'])
def test_maxpool_with_a_skip_consumer_sums_both_gradients_in_one_pass(shape, use):
    """U-Net skip connection (reference Module.py:116-132): the encoder feature feeds MaxPool2d and the decoder.  ``ops.maxpool2_skip``
    returns (x, pooled) from ONE node whose backward writes skip gradient + routed pooled gradient in a single kernel
    (fcd_maxpool2_bwd_add); the result must equal autograd's accumulation over ``maxpool2(x)`` and x bit for bit, odd trailing
    rows / columns and unused outputs included."""
    ops = _ops()
    x0 = rnd(*shape, seed=111)
    gs = rnd(*shape, seed=112).cuda()
    gp = rnd(shape[0], shape[1], shape[2] // 2, shape[3] // 2, seed=113).cuda()
    out = {}
    for tag in ('fused', 'plain'):
        x = x0.cuda().requires_grad_(True)
        h = x * 1.0
        if tag == 'fused':
            skip, pooled = ops.maxpool2_skip(h)
        else:
            skip, pooled = h, ops.maxpool2(h)
        loss = 0.0
        if use != 'pool_only':
            loss = loss + (skip * gs).sum()
        if use != 'skip_only':
            loss = loss + (pooled * gp).sum()
        loss.backward()
        out[tag] = (x.grad.cpu(), skip.detach().cpu(), pooled.detach().cpu())
    for a, r, what in zip(out['fused'], out['plain'], ('dx', 'skip', 'pooled')):
        assert torch.equal(a, r), what
    assert torch.equal(out['fused'][1], x0)


@pytest.mark.parametrize('case', [(2, 128, 64, 64, 1), (4, 32, 32, 32, 2), (3, 21, 32, 36, 1), (8, 128, 128, 128, 1)], ids=lambda c: 'x'.join(map(str, c)))
def test_head_behind_batchnorm_relu_without_the_activation_tensor(case):
    """BatchNorm2d(train) -> ReLU -> OutConv (reference Module.py:25-31 -> :82-90): ``ops.bn_relu_head`` (normalisation, ReLU, 1x1
    filter, sigmoid in one pass over the BatchNorm input; backward = two passes) against ``conv1x1_head(bn_act(z))``.  The forward
    kernels form the same products in the same order: bit-identical y and running statistics.  The gradients sum w[c] * g once per
    channel instead of per element (fp64 sums): 2e-6 of the largest entry."""
    ops = _ops()
    N, C, H, W, G = case
    z0 = rnd(N, C, H, W, seed=121)
    w = rnd(1, C, 1, 1, seed=122, scale=C ** -0.5)
    b = rnd(1, seed=123, scale=0.1)
    g = rnd(N, 1, H, W, seed=124)
    res = {}
    for tag in ('fused', 'separate'):
        bn = torch.nn.BatchNorm2d(C).cuda().train()
        with torch.no_grad():
            bn.weight.copy_(rnd(C, seed=125).cuda() * 0.2 + 1.0)
            bn.bias.copy_(rnd(C, seed=126).cuda() * 0.1)
        zin = z0.cuda().requires_grad_(True)
        wg, bg = (t.cuda().requires_grad_(True) for t in (w, b))
        z = zin * 1.0
        if tag == 'fused':
            assert ops.bn_relu_head_ok(z, bn, wg, G)
            y = ops.bn_relu_head(z, bn, wg, bg, sigmoid=True, groups=G)
        else:
            y = ops.conv1x1_head(ops.bn_act(z, bn, ops.ACT_RELU, groups=G), wg, bg, sigmoid=True)
        y.backward(g.cuda())
        res[tag] = [t.detach().cpu() for t in (y, bn.running_mean, bn.running_var, zin.grad, wg.grad, bg.grad, bn.weight.grad, bn.bias.grad)]
        assert int(bn.num_batches_tracked) == G
    names = ('y', 'running_mean', 'running_var', 'dz', 'dw', 'db', 'dgamma', 'dbeta')
    for a, r, what in zip(res['fused'][:3], res['separate'][:3], names[:3]):
        assert torch.equal(a, r), what
    for a, r, what in zip(res['fused'][3:], res['separate'][3:], names[3:]):
        err = (a.double() - r.double()).abs().max().item() / max(r.double().abs().max().item(), 1e-30)
        assert err < 2e-6, (what, err)


@pytest.mark.parametrize('case', [(4, 16, 16, 24, 2), (2, 64, 64, 64, 1), (6, 8, 10, 12, 3), (16, 64, 128, 128, 2)], ids=lambda c: 'x'.join(map(str, c)))
@pytest.mark.parametrize('use', ['both', 'pool_only', 'skip_only'])
def test_encoder_tail_batchnorm_relu_maxpool_skip_as_one_node(case, use):
    """BatchNorm2d(train) -> ReLU -> {MaxPool2d(2), skip connection} (reference Module.py:30-31 -> :43-44, :116-132):
    ``ops.bn_relu_pool_skip`` (activation and pooled tensor from one read of z; backward recomputes argmax and ReLU gate from z and
    never touches the activation or the summed gradient) against ``a = bn_act(z); (a, maxpool2(a))`` with autograd's accumulation.
    Activation, pooled tensor and running statistics are bit-identical (same fma, same window order); the input / affine
    gradients differ by the order of the fp64 channel sums only."""
    ops = _ops()
    N, C, H, W, G = case
    z0 = rnd(N, C, H, W, seed=131)
    gs = rnd(N, C, H, W, seed=132).cuda()
    gp = rnd(N, C, H // 2, W // 2, seed=133).cuda()
    res = {}
    for tag in ('fused', 'separate'):
        bn = torch.nn.BatchNorm2d(C).cuda().train()
        with torch.no_grad():
            bn.weight.copy_(rnd(C, seed=134).cuda() * 0.2 + 1.0)
            bn.bias.copy_(rnd(C, seed=135).cuda() * 0.1)
        zin = z0.cuda().requires_grad_(True)
        z = zin * 1.0
        if tag == 'fused':
            assert ops.bn_relu_pool_skip_ok(z, bn, G)
            a, p = ops.bn_relu_pool_skip(z, bn, groups=G)
        else:
            a = ops.bn_act(z, bn, ops.ACT_RELU, groups=G)
            p = ops.maxpool2(a)
        loss = 0.0
        if use != 'pool_only':
            loss = loss + (a * gs).sum()
        if use != 'skip_only':
            loss = loss + (p * gp).sum()
        loss.backward()
        res[tag] = [t.detach().cpu() for t in (a, p, bn.running_mean, bn.running_var, zin.grad, bn.weight.grad, bn.bias.grad)]
        assert int(bn.num_batches_tracked) == G
    names = ('a', 'pooled', 'running_mean', 'running_var', 'dz', 'dgamma', 'dbeta')
    for x, r, what in zip(res['fused'][:4], res['separate'][:4], names[:4]):
        assert torch.equal(x, r), what
    for x, r, what in zip(res['fused'][4:], res['separate'][4:], names[4:]):
        err = (x.double() - r.double()).abs().max().item() / max(r.double().abs().max().item(), 1e-30)
        assert err < 2e-6, (what, err)


def _split_modes_conv(ops, x, w, b):
    """y of the 3x3 layer with the F(4x4) GEMMs on the fp32 matrix pipe (mode 0) and on the two split-bf16 kernels."""
    lib = ops.lib
    prev = lib.fcd_conv_wino_split_set(-1)
    out = {}
    try:
        for mode in (0, 1, 2):
            lib.fcd_conv_wino_split_set(mode)
            with torch.no_grad():
                out[mode] = ops.conv2d(x.cuda(), w.cuda(), b.cuda(), 1, 1).cpu()
    finally:
        lib.fcd_conv_wino_split_set(prev)
    return out


@pytest.mark.parametrize('scale', ['mixed_1e+-12', 'big_1e30', 'small_1e-30', 'tiny_1e-36'])
def test_wino_split_gemm_operand_magnitudes(scale):
    """the exact three-way bf16 split on the GPU with operands far from N(0,1) -- per-channel
    magnitudes spread over 24 decades (every reduction mixes them), |x| ~ 1e30 against |w| ~ 1e-30, and the other way
    round, and activations at 1e-36 where the LOW parts of the split (2^-16 x) are fp32 subnormals.  Truth = torch fp64;
    the split path must stay within the fp32 matrix pipe's error (both carry the F(4x4) transforms' rounding)."""
    ops = _ops()
    N, C, H, W, K = 2, 256, 32, 32, 256
    x = rnd(N, C, H, W, seed=61)
    w = rnd(K, C, 3, 3, seed=62, scale=(2.0 / (C * 9)) ** 0.5)
    b = torch.zeros(K)
    if scale == 'mixed_1e+-12':
        e = torch.linspace(-12, 12, C)
        x = x * (10.0 ** e).view(1, C, 1, 1).float()
        w = w * (10.0 ** (-e)).view(1, C, 1, 1).float()
    elif scale == 'big_1e30':
        x, w = x * 1e30, w * 1e-30
    elif scale == 'small_1e-30':
        x, w = x * 1e-30, w * 1e30
    else:
        x, w = x * 1e-36, w * 1e36
    yr = F.conv2d(x.double(), w.double(), None, padding=1)
    ys = yr.abs().max().item()
    out = _split_modes_conv(ops, x, w, b)
    err = {m: ((out[m].double() - yr).abs().max().item() / ys, (out[m].double() - yr).pow(2).mean().sqrt().item() / ys) for m in out}
    print('\n[split GEMM, %s] (max, rms) error / max|y|: fp32 pipe %.2e/%.2e  split128 %.2e/%.2e  split256 %.2e/%.2e'
          % ((scale,) + err[0] + err[1] + err[2]))
    for m in (0, 1, 2):
        assert torch.isfinite(out[m]).all(), (scale, m)
    if scale == 'tiny_1e-36':
        # low parts of the activations' split are fp32 subnormals (2^-16 * 1e-36 < 1.2e-38): whatever the matrix pipe
        # does with subnormal bf16 inputs, the result keeps at least the two upper parts (2^-16 relative) -- and the
        # fp32 pipe on the same data is held to the same bound, so the difference between them is on record
        for m in (0, 1, 2):
            assert err[m][0] < 2e-4, (scale, m, err[m])
        return
    for m in (0, 1, 2):
        assert err[m][0] < 6e-5, (scale, m, err[m])
    for m in (1, 2):
        assert err[m][1] <= 1.15 * err[0][1] + 1e-8, (scale, m, err)


def test_wino_split_gemm_nonfinite_propagation():
    """Inf / NaN in the activations.  Winograd itself spreads a non-finite pixel over every output tile whose 6x6 input
    patch holds it (B^T d B and A^T M A form differences: Inf - Inf = NaN) -- on ANY matrix pipe -- where the direct
    kernel confines it to the 3x3 neighbourhood.  The split adds one more Inf - Inf (x - bf16(x)), so an Inf becomes NaN
    where the fp32 pipe may keep +-Inf.  Checked here: (1) every output the direct convolution
    makes non-finite is non-finite on both pipes; (2) the split path flags every output the fp32 pipe flags; (3) nothing
    leaks outside the tiles whose patch holds the bad pixel, and there the finite results are the usual ones."""
    ops = _ops()
    N, C, H, W, K = 1, 128, 32, 32, 128
    x = rnd(N, C, H, W, seed=71)
    w = rnd(K, C, 3, 3, seed=72, scale=(2.0 / (C * 9)) ** 0.5)
    b = torch.zeros(K)
    clean = F.conv2d(x.double(), w.double(), None, padding=1)
    x = x.clone()
    x[0, 5, 9, 13] = float('inf')
    x[0, 77, 22, 6] = float('nan')
    direct_bad = ~torch.isfinite(F.conv2d(x, w, None, padding=1)).any(dim=1)          # (N, H, W)
    out = _split_modes_conv(ops, x, w, b)
    tiles_bad = torch.zeros(N, H, W, dtype=torch.bool)
    for (pi, pj) in ((9, 13), (22, 6)):
        for ti in range(H // 4):
            for tj in range(W // 4):
                if 4 * ti - 1 <= pi <= 4 * ti + 4 and 4 * tj - 1 <= pj <= 4 * tj + 4:
                    tiles_bad[0, 4 * ti:4 * ti + 4, 4 * tj:4 * tj + 4] = True
    bad = {m: ~torch.isfinite(out[m]).all(dim=1) for m in out}
    for m in (0, 1, 2):
        assert (bad[m] | ~direct_bad).all(), ('mode %d lost a non-finite output of the direct convolution' % m)
        assert (~bad[m] | tiles_bad).all(), ('mode %d: non-finite values outside the tiles that hold the bad pixels' % m)
        ok = ~tiles_bad
        e = ((out[m].double() - clean).abs().amax(dim=1)[ok]).max().item() / clean.abs().max().item()
        assert e < 6e-5, (m, e)
    for m in (1, 2):
        assert (bad[m] | ~bad[0]).all(), 'the split path must flag every output the fp32 pipe flags'
    n_inf = {m: int(torch.isinf(out[m]).sum()) for m in out}
    print('\n[non-finite propagation] flagged pixels: direct %d, fp32 pipe %d, split %d / %d (of %d in the touched tiles); '
          '+-Inf values kept: %s' % (int(direct_bad.sum()), int(bad[0].sum()), int(bad[1].sum()), int(bad[2].sum()),
                                      int(tiles_bad.sum()), n_inf))


@pytest.mark.parametrize('case', [(2, 128, 32, 32, 256, 256), (1, 64, 40, 72, 128, 128), (3, 256, 16, 32, 512, 384)],
                         ids=lambda c: 'x'.join(map(str, c)))
@pytest.mark.parametrize('relu', [False, True])
def test_conv3x3_pair_cat_equals_conv_of_concatenation(case, relu):
    """Decoder convolution reading [branch-1 skip | branch-2 skip | upsampled] in place (ops.conv3x3_pair_cat) vs the same
    layer on torch.cat([...], dim=1): outputs, both input gradients, filter and bias gradients -- bit for bit (same
    kernels, same summation order; only the addressing differs)."""
    ops = _ops()
    n, c, H, W, cu, K = case
    f = rnd(2 * n, c, H, W, seed=41).cuda()
    u = rnd(n, cu, H, W, seed=42).cuda()
    w = rnd(K, 2 * c + cu, 3, 3, seed=43, scale=(2.0 / ((2 * c + cu) * 9)) ** 0.5).cuda()
    b = rnd(K, seed=44, scale=0.1).cuda()
    g = rnd(n, K, H, W, seed=45).cuda()
    if ops.lib.fcd_conv_wino_set(-1) == 0:
        pytest.skip('direct kernels only (FCD_WINO=0): the tensor-list convolution is an F(4x4) path')
    assert ops.conv3x3_pair_cat_ok(f, u, w)
    f1, u1, w1, b1 = (t.clone().requires_grad_(True) for t in (f, u, w, b))
    y1 = ops.conv3x3_pair_cat(f1, u1, w1, b1, relu=relu)
    y1.backward(g)
    f2, u2, w2, b2 = (t.clone().requires_grad_(True) for t in (f, u, w, b))
    y2 = ops.conv2d(torch.cat([f2[:n], f2[n:], u2], dim=1), w2, b2, 1, 1, relu=relu)
    y2.backward(g)
    assert torch.equal(y1, y2)
    assert torch.equal(f1.grad, f2.grad) and torch.equal(u1.grad, u2.grad)
    assert torch.equal(w1.grad, w2.grad) and torch.equal(b1.grad, b2.grad)
    # layers off the F(4x4) path are refused (the caller concatenates)
    assert not ops.conv3x3_pair_cat_ok(f[:, :16], u, w[:, :32 + cu])
    assert not ops.conv3x3_pair_cat_ok(f[..., :2, :], u[..., :2, :], w)


def test_conv_winograd_matches_direct_kernels():
    """Same layer through the direct MFMA kernel and both Winograd tile sizes."""
    ops = _ops()
    lib = ops.lib
    x = rnd(2, 512, 24, 40, seed=21).cuda()
    w = rnd(256, 512, 3, 3, seed=22, scale=(2.0 / (512 * 9)) ** 0.5).cuda()
    b = rnd(256, seed=23, scale=0.1).cuda()
    outs = {}
    prev = lib.fcd_conv_wino_set(0)
    try:
        for m in (0, 2, 4):
            lib.fcd_conv_wino_set(m)
            with torch.no_grad():
                outs[m] = ops.conv2d(x, w, b, 1, 1)
    finally:
        lib.fcd_conv_wino_set(prev)
    sc = outs[0].abs().max().item()
    assert (outs[2] - outs[0]).abs().max().item() <= 5e-6 * sc
    assert (outs[4] - outs[0]).abs().max().item() <= 6e-5 * sc


@pytest.mark.parametrize('case', [(2, 1, 36, 68, 64), (1, 3, 17, 132, 72), (3, 4, 16, 64, 40), (2, 1, 21, 30, 64),
                                  (2, 1, 37, 64, 64), (1, 1, 16, 128, 64), (2, 1, 50, 256, 64)],      # one band, 64 filters, W % 64 == 0: MFMA data gradient
                         ids=lambda c: 'x'.join(map(str, c)))
def test_thin_conv_relu_bitmask_path(case):
    """Frozen thin-channel layer (VGG conv1_1 on single bands): conv + bias + ReLU whose backward mask is kept as
    4 bits per 1 x 4 strip (`fcd_conv2d_fwd_relu_bits` / `_bwd_data_bits`; W % 4 == 0 and K % 8 == 0, else the
    fp32-mask path) vs the ATen composition."""
    import ctypes
    ops = _ops()
    N, C, H, W, K = case
    x = rnd(N, C, H, W, seed=201)
    w = rnd(K, C, 3, 3, seed=202, scale=(2.0 / (C * 9)) ** 0.5)
    b = rnd(K, seed=203, scale=0.1)
    g = rnd(N, K, H, W, seed=204)
    xr = x.clone().requires_grad_(True)
    yr = F.relu(F.conv2d(xr, w, b, padding=1))
    yr.backward(g)
    xg = x.cuda().requires_grad_(True)
    d = ops._desc(x.shape, w.shape, 1, 1)
    expect_bits = W % 4 == 0 and K % 8 == 0
    assert (ops.lib.fcd_conv2d_relu_bits_bytes(ctypes.byref(d)) > 0) == expect_bits
    y = ops.conv2d(xg, w.cuda(), b.cuda(), 1, 1, relu=True)          # frozen filter => bit-mask path when available
    y.backward(g.cuda())
    assert_close(y, yr, what='y')
    dd = xg.grad.cpu().double() - xr.grad.double()
    assert (dd.norm() / xr.grad.double().norm()).item() < 1e-4, 'dx'       # kink flips only
    assert (dd.abs() > 1e-4 * xr.grad.abs().max().item()).double().mean().item() < 2e-3


W2_CASES = [
    # N, C, H, W, K   (3x3 / stride 1 / pad 1; 33..64 GEMM rows in at least one direction, >= 32 reduction channels)
    (2, 64, 40, 64, 64),       # forward and data gradient on the fused kernel
    (1, 64, 13, 27, 64),       # odd sizes: ragged 8 x 32 blocks, odd pooled extents
    (3, 40, 9, 70, 48),        # 48 rows (padded to 64), 40 reduction channels, three column blocks
    (2, 128, 20, 28, 64),      # forward fused (64 rows), data gradient on the three-kernel / direct path (128 rows)
    (2, 64, 16, 36, 128),      # forward NOT fused (128 rows), data gradient fused (64 rows, 128 reduction channels)
    (1, 34, 6, 5, 64),         # channel tail (34 = 8 chunks + 2), tiny map
]


@pytest.mark.parametrize('case', W2_CASES, ids=lambda c: 'x'.join(map(str, c)))
def test_conv_wino2_fused_kernel(case):
    """Fused Winograd F(2x2,3x3) kernel (csrc/conv_wino2.hip) vs torch fp64: forward with every epilogue (bias, ReLU,
    PReLU + residual, ReLU + max-pool + code), data gradient with every source (plain, ReLU-gated, pooled-gradient
    routed), weight gradient unchanged.  F(2x2) transforms only add / halve: error stays at direct-kernel level."""
    import ctypes
    ops = _ops()
    lib = ops.lib
    N, C, H, W, K = case
    d = ops._desc((N, C, H, W), (K, C, 3, 3), 1, 1)
    fused_f, fused_d = bool(lib.fcd_conv_wino2_plan(ctypes.byref(d), 0)), bool(lib.fcd_conv_wino2_plan(ctypes.byref(d), 1))
    assert fused_f == (32 < K <= 64 and C >= 32) and fused_d == (32 < C <= 64 and K >= 32)
    assert fused_f or fused_d
    tol, flip_tol = 5e-6, 4e-3
    tol_f = tol if fused_f else 6e-5        # a forward with > 64 rows runs on the three-kernel F(4x4) path
    if not (fused_f and fused_d):
        flip_tol = 2e-2
    x = rnd(N, C, H, W, seed=31)
    w = rnd(K, C, 3, 3, seed=32, scale=(2.0 / (C * 9)) ** 0.5)
    b = rnd(K, seed=33, scale=0.1)
    g = rnd(N, K, H, W, seed=34)
    xr, wr, br = (t.double().requires_grad_(True) for t in (x, w, b))
    yr = F.conv2d(xr, wr, br, padding=1)
    yr.backward(g.double())
    xg, wg, bg = (t.cuda().requires_grad_(True) for t in (x, w, b))
    y = ops.conv2d(xg, wg, bg, 1, 1)
    y.backward(g.cuda())
    assert_close(y, yr.float(), tol=tol_f, what='y')
    # (a data gradient with > 64 GEMM rows runs on the three-kernel F(4x4) path: ~1e-5 transform rounding)
    assert_close(xg.grad, xr.grad.float(), tol=tol if fused_d else 6e-5, what='dx')
    assert_close(wg.grad, wr.grad.float(), tol=5e-5, what='dw')
    assert torch.equal(ops.conv2d(xg, wg, bg, 1, 1), y)                     # bit-reproducible
    # the direct kernels give the same numbers to rounding (A/B switch of the library)
    prev = lib.fcd_conv_wino_set(0)
    try:
        assert not lib.fcd_conv_wino2_plan(ctypes.byref(d), 0)
        with torch.no_grad():
            y_direct = ops.conv2d(xg, wg, bg, 1, 1)
    finally:
        lib.fcd_conv_wino_set(prev)
    assert (y_direct - y).abs().max().item() <= 2 * tol_f * y.abs().max().item()
    # ReLU epilogue + mask-gated data gradient
    xr3 = x.double().requires_grad_(True)
    F.relu(F.conv2d(xr3, w.double(), b.double(), padding=1)).backward(g.double())
    xg3 = x.cuda().requires_grad_(True)
    y3 = ops.conv2d(xg3, w.cuda(), b.cuda(), 1, 1, relu=True)
    y3.backward(g.cuda())
    assert_close(y3, F.relu(yr).float(), tol=tol_f, what='relu y')
    dd = xg3.grad.cpu().double() - xr3.grad
    assert (dd.norm() / xr3.grad.norm()).item() < flip_tol, 'relu dx'
    # ReLU + max-pool epilogue (ties: constant block; all-negative channels) + code-routed data gradient
    if H >= 2 and W >= 2 and K > 32 and C > 32:
        x2 = x.clone()
        x2[:, :, : H // 2, : W // 2] = x2[:, :, :1, :1]
        b2 = b.clone()
        b2[: K // 4] = -50.0
        xr2 = x2.double().requires_grad_(True)
        ypr = F.max_pool2d(F.relu(F.conv2d(xr2, w.double(), b2.double(), padding=1)), 2)
        gp = rnd(*ypr.shape, seed=35)
        ypr.backward(gp.double())
        xg2 = x2.cuda().requires_grad_(True)
        yp = ops.conv2d_relu_maxpool2(xg2, w.cuda(), b2.cuda())
        yp.backward(gp.cuda())
        assert_close(yp, ypr.float(), tol=tol_f, what='pooled y')
        dd = xg2.grad.cpu().double() - xr2.grad
        assert (dd.norm() / xr2.grad.norm().clamp_min(1e-30)).item() < flip_tol, 'pooled dx'
    # inference epilogue: PReLU + residual
    res = rnd(N, K, H, W, seed=36)
    slope = torch.tensor([0.25])
    got = ops.conv2d_infer(x.cuda(), w.cuda(), b.cuda(), 1, 1, ops.ACT_PRELU, slope=slope.cuda(), residual=res.cuda())
    ref = yr.detach()
    ref = torch.where(ref > 0, ref, ref * 0.25) + res.double()
    assert_close(got, ref.float(), tol=tol_f, what='prelu + residual')


@pytest.mark.parametrize('case', [(2, 128, 64, 64, 128), (3, 256, 32, 48, 128), (2, 128, 30, 44, 256)], ids=lambda c: 'x'.join(map(str, c)))
def test_weight_gradient_from_the_forward_v_equals_the_one_from_x(case):
    """``fcd_conv2d_bwd_weight_bias_v`` (the weight gradient as a GEMM over the transformed input V the FORWARD pass
    left behind, ``fcd_conv2d_fwd_wino_keepv``) against ``fcd_conv2d_bwd_weight_bias`` (which transforms x again) on the same
    layer: identical transform values, the B operand only read in another layout -- equal within fp32 round-off, and both
    within the Winograd weight-gradient tolerance of the fp64 gradient."""
    import ctypes
    ops = _ops()
    lib = ops.lib
    N, C, H, W, K = case
    d = ops._desc((N, C, H, W), (K, C, 3, 3), 1, 1)
    nb = lib.fcd_conv_wino_keepv_bytes(ctypes.byref(d))
    if not nb:
        pytest.skip('this layer does not keep V')
    x, w, b = rnd(N, C, H, W, seed=71).cuda(), rnd(K, C, 3, 3, seed=72, scale=(2.0 / (C * 9)) ** 0.5).cuda(), rnd(K, seed=73).cuda()
    dy = rnd(N, K, H, W, seed=74).cuda()
    y = torch.empty((N, K, H, W), device='cuda')
    vk = torch.empty(nb // 4, device='cuda')
    ws = ops._ws(lib.fcd_conv_wino_ws_bytes(ctypes.byref(d), 0), x.device)
    ops.check(lib.fcd_conv2d_fwd_wino_keepv(ctypes.byref(d), ops._p(x), ops._p(ops.wino_weight(w, 0, 4)), ops._p(b), ops._p(y), 0, None,
                                            None, ops._p(ws), ws.numel(), ops._p(vk), ops._stream()), 'fwd_wino_keepv')
    out = {}
    for tag in ('v', 'x'):
        dw, db = torch.zeros_like(w), torch.zeros_like(b)
        ws = ops._ws(lib.fcd_conv2d_bwd_weight_ws_bytes(ctypes.byref(d)), x.device)
        if tag == 'v':
            ops.check(lib.fcd_conv2d_bwd_weight_bias_v(ctypes.byref(d), ops._p(vk), ops._p(dy), None, ops._p(dw), ops._p(db), ops._p(ws),
                                                       ws.numel(), ops._stream()), 'bwd_weight_bias_v')
        else:
            ops.check(lib.fcd_conv2d_bwd_weight_bias(ctypes.byref(d), ops._p(x), ops._p(dy), None, ops._p(dw), ops._p(db), ops._p(ws),
                                                     ws.numel(), ops._stream()), 'bwd_weight_bias')
        out[tag] = (dw.cpu().double(), db.cpu().double())
    scale = out['x'][0].abs().max().item()
    assert (out['v'][0] - out['x'][0]).abs().max().item() <= 2e-6 * scale
    assert torch.equal(out['v'][1], out['x'][1])                                  # the bias gradient is the same dY pass
    xr, wr = x.cpu().double(), w.cpu().double().requires_grad_(True)
    F.conv2d(xr, wr, b.cpu().double(), padding=1).backward(dy.cpu().double())
    assert_close(out['v'][0], wr.grad, tol=6e-5, what='dw from V')
    assert_close(out['v'][1], dy.cpu().double().sum(dim=(0, 2, 3)), tol=2e-5, what='db')


@pytest.mark.parametrize('case', [(2, 64, 40, 64, 64), (3, 64, 17, 96, 128), (2, 128, 24, 32, 64), (1, 40, 20, 64, 72), (5, 64, 9, 128, 64),
                                  # stride 2 (the Discriminator's layers; bf16 pipe only): square, odd height, ragged last tile + ragged filter count, one short tile
                                  (2, 64, 32, 32, 128, 2), (3, 64, 17, 48, 64, 2), (2, 128, 24, 40, 72, 2), (1, 256, 16, 16, 512, 2), (4, 64, 64, 128, 128, 2)],
                         ids=lambda c: 'x'.join(map(str, c)))
def test_weight_gradient_on_nchw_operands(case, switches):
    """The rolling 3x3 weight-gradient kernel that reads x and dy as they are (``conv_wgrad_roll_nchw_kernel``, W % 32 == 0, no ReLU
    mask; reference: autograd of nn.Conv2d, Module.py:25-31,177-181) against the round-1..4 route (channel-minor copies of both operands +
    ``conv_wgrad_roll_kernel``, switch WGRAD_NCHW=0) and against the fp64 gradient: several column strips, an odd row count, two filter
    tiles, two channel tiles, ragged channel / filter counts (zero page), more samples than splits."""
    import ctypes
    ops = _ops()
    lib = ops.lib
    N, C, H, W, K = case[:5]
    st = case[5] if len(case) > 5 else 1
    d = ops._desc((N, C, H, W), (K, C, 3, 3), st, 1)
    P, Q = (H - 1) // st + 1, (W - 1) // st + 1
    x, dy = rnd(N, C, H, W, seed=171).cuda(), rnd(N, K, P, Q, seed=174).cuda()
    out = {}
    for tag in (('nchw', 'nchw_fp32_pipe', 'copies') if st == 1 else ('nchw', 'copies')):
        switches('WGRAD_NCHW', 0 if tag == 'copies' else 1)
        switches('WGRAD_SPLIT', 0 if tag == 'nchw_fp32_pipe' else 1)     # default: bf16 pipe, operands split exactly in three
        dw, db = torch.full((K, C, 3, 3), float('nan'), device='cuda'), torch.full((K,), float('nan'), device='cuda')
        ws = ops._ws(lib.fcd_conv2d_bwd_weight_ws_bytes(ctypes.byref(d)), x.device)
        ops.check(lib.fcd_conv2d_bwd_weight_bias(ctypes.byref(d), ops._p(x), ops._p(dy), None, ops._p(dw), ops._p(db), ops._p(ws),
                                                 ws.numel(), ops._stream()), 'bwd_weight_bias')
        out[tag] = (dw.cpu().double(), db.cpu().double())
    xr, wr = x.cpu().double(), torch.zeros(K, C, 3, 3, dtype=torch.double, requires_grad=True)
    F.conv2d(xr, wr, None, stride=st, padding=1).backward(dy.cpu().double())
    for tag in out:
        assert_close(out[tag][0], wr.grad, tol=2e-5, what='dw (%s)' % tag)
        assert_close(out[tag][1], dy.cpu().double().sum(dim=(0, 2, 3)), tol=2e-5, what='db (%s)' % tag)
    scale = wr.grad.abs().max().item()
    for tag in [t for t in out if t != 'copies']:
        assert (out[tag][0] - out['copies'][0]).abs().max().item() <= 4e-6 * scale, tag      # same products (the split ones: up to 2^-24 each), another summation order


@pytest.mark.parametrize('shape', [(2, 2, 512, 16, 16), (1, 3, 40, 7, 5), (2, 1, 64, 13, 13)], ids=lambda c: 'x'.join(map(str, c)))
def test_pair_gap_diff(shape):
    """AdaptiveAvgPool2d(1)(net(x) - net(y)) of the Discriminator (reference Module.py:211,222-223) on the batched feature
    tensor: one kernel, element-wise difference first, fp64 accumulator -- against the same expression in fp64."""
    ops = _ops()
    pairs, n, C, H, W = shape
    f = rnd(2 * pairs * n, C, H, W, seed=81)
    f[n:2 * n] = f[:n] + 1e-3 * rnd(n, C, H, W, seed=82)          # nearly equal branches: the cancelling case
    fg = f.cuda().requires_grad_(True)
    d = ops.pair_gap_diff(fg, pairs)
    g = rnd(pairs * n, C, 1, 1, seed=83)
    d.backward(g.cuda())
    fr = f.double().requires_grad_(True)
    fr5 = fr.view(pairs, 2, n, C, H, W)
    dr = (fr5[:, 0] - fr5[:, 1]).mean(dim=(3, 4)).reshape(pairs * n, C, 1, 1)
    dr.backward(g.double())
    assert d.shape == (pairs * n, C, 1, 1)
    # exact to fp32 rounding of the RESULT (the fp32 differences are the only rounded intermediates)
    err = (d.detach().cpu().double() - dr.detach()).abs().max().item()
    assert err <= 1.2e-7 * dr.detach().abs().max().item() + 1.2e-7 * f.abs().max().item() * 2 ** -3, err
    assert_close(fg.grad, fr.grad, tol=1e-6, what='df')


@pytest.mark.parametrize('case', [(4, 2, 13, 32, 32), (2, 3, 3, 7, 5), (1, 1, 5, 16, 8), (3, 2, 4, 9, 9)], ids=lambda c: 'x'.join(map(str, c)))
def test_masked_stack(case):
    """cat([t * (1 - cmask) for t in tensors]) -- the reference's `x * (1 - cmask).repeat(1, C, 1, 1)` in front of the
    Discriminator / perception VGG / SSIM (Demo_RSSS.py:290-300, Loss.py:78-79,111-112) -- as one kernel: forward bit-equal to
    the ATen sequence, backward (source gradients where wanted, mask gradient with an fp64 accumulator) against fp64 autograd.
    The same tensor may stand in two slots (x of both Discriminator pairs)."""
    ops = _ops()
    k, N, C, H, W = case
    ts = [rnd(N, C, H, W, seed=90 + i) for i in range(k)]
    if k >= 3:
        ts[2] = ts[0]                                   # duplicate slot
    cm = torch.sigmoid(rnd(N, 1, H, W, seed=99))
    want_src = [i % 2 == 1 for i in range(k)]           # odd slots want a gradient
    tg = []
    for i, t in enumerate(ts):
        if k >= 3 and i == 2:
            tg.append(tg[0])
        else:
            tg.append(t.cuda().requires_grad_(want_src[i]))
    cg = cm.cuda().requires_grad_(True)
    z = ops.masked_stack(tg, cg)
    ref = torch.cat([t.cuda() * (1 - cm.cuda()) for t in ts], dim=0)
    assert z.shape == (k * N, C, H, W)
    assert torch.equal(z.detach(), ref)
    g = rnd(k * N, C, H, W, seed=77)
    z.backward(g.cuda())
    td = [t.double().requires_grad_(True) for t in ts]
    cd = cm.double().requires_grad_(True)
    zr = torch.cat([t * (1 - cd) for t in td], dim=0)
    zr.backward(g.double())
    assert_close(cg.grad, cd.grad, tol=2e-6, what='dcmask')
    for i in range(k):
        if k >= 3 and i in (0, 2):
            continue
        if want_src[i]:
            assert_close(tg[i].grad, td[i].grad, tol=1e-6, what='dsrc%d' % i)
        else:
            assert tg[i].grad is None
    # no mask gradient wanted: only the source gradients are written
    t1 = ts[0].cuda().requires_grad_(True)
    z2 = ops.masked_stack([t1], cm.cuda())
    z2.backward(g[:N].cuda())
    assert_close(t1.grad, (g[:N].double() * (1 - cm.double())), tol=1e-6, what='dsrc only')
    with pytest.raises(ValueError):
        ops.masked_stack([ts[0].cuda(), ts[0].cuda()[:, :1]], cm.cuda())


def _misaligned(t):
    """The same values as a CONTIGUOUS view whose storage starts 4 bytes past a 16-B boundary."""
    buf = torch.empty(t.numel() + 1, dtype=t.dtype, device=t.device)
    v = buf[1:].view(t.shape)
    v.copy_(t)
    assert v.is_contiguous() and v.data_ptr() % 16 == 4
    return v


def test_vector_paths_fall_back_on_misaligned_views():
    """the float4 paths of masked_stack / pair_gap_diff / BatchNorm were chosen from HW % 4 == 0 alone; a contiguous
    view with a storage offset is not 16-B aligned.  Same results from aligned tensors and from views one float off."""
    ops = _ops()
    N, C, H, W = 2, 8, 16, 16
    x = rnd(N, C, H, W, seed=1).cuda()
    y = rnd(N, C, H, W, seed=2).cuda()
    cm = torch.sigmoid(rnd(N, 1, H, W, seed=3)).cuda()
    g = rnd(2 * N, C, H, W, seed=4).cuda()

    def stack(xx, yy, mm, gg):
        xx, yy, mm = xx.detach().requires_grad_(True), yy.detach().requires_grad_(True), mm.detach().requires_grad_(True)
        z = ops.masked_stack([xx, yy], mm)
        z.backward(gg)
        return z.detach(), xx.grad, yy.grad, mm.grad
    for u, v in zip(stack(x, y, cm, g), stack(_misaligned(x), _misaligned(y), _misaligned(cm), _misaligned(g))):
        assert torch.equal(u, v)

    f = rnd(4 * N, C, H, W, seed=5).cuda()
    gd = rnd(2 * N, C, 1, 1, seed=6).cuda()

    def gap(ff):
        ff = ff.detach().requires_grad_(True)
        d = ops.pair_gap_diff(ff, 2)
        d.backward(gd)
        return d.detach(), ff.grad
    for u, v in zip(gap(f), gap(_misaligned(f))):
        assert torch.equal(u, v)

    bn = torch.nn.BatchNorm2d(C).cuda().train()

    def bnrun(xx, gg):
        bn.running_mean.zero_(); bn.running_var.fill_(1.0)
        xx = xx.detach().requires_grad_(True)
        z = ops.bn_act(xx, bn, ops.ACT_RELU)
        z.backward(gg)
        return z.detach(), xx.grad, bn.weight.grad.clone(), bn.running_var.clone()
    ga = g[:N].contiguous()
    a = bnrun(x, ga)
    bn.weight.grad = None; bn.bias.grad = None
    b = bnrun(_misaligned(x), _misaligned(ga))
    for u, v in zip(a, b):
        assert torch.equal(u, v)


def test_wino_packs_refreshed_in_one_launch_are_bit_identical():
    """ops.refresh_packs: after an optimizer step every F(4x4) filter pack of the stepped parameters is re-packed in place by ONE
    launch (fcd_conv_wino_pack_multi) -- same bytes as the per-layer fcd_conv_wino_pack calls, cache current, other packs dropped."""
    import fcd_gan_pytorch_amd as p
    ops = _ops()
    convs = [torch.nn.Conv2d(128, 256, 3, padding=1), torch.nn.Conv2d(256, 128, 3, padding=1), torch.nn.Conv2d(64, 64, 3, padding=1),
             torch.nn.Conv2d(128, 128, 3, padding=1)]
    net = torch.nn.Sequential(*convs).cuda()
    opt = p.optim.RMSprop(net.parameters(), lr=1e-3)
    x = rnd(2, 128, 32, 32, seed=1).cuda()

    def packs():
        out = {}
        for i, c in enumerate(convs):
            for key, (ver, buf) in c.weight.__dict__.get('_fcd_pack', {}).items():
                out[(i, key)] = (ver, buf.data_ptr(), buf.clone())
        return out

    def fwd_bwd():
        h = ops.conv2d(x, convs[0].weight, convs[0].bias, 1, 1, relu=False)
        h = ops.conv2d(h, convs[1].weight, convs[1].bias, 1, 1, relu=False)
        g = ops.conv2d(h[:, :64].contiguous(), convs[2].weight, convs[2].bias, 1, 1)      # 64 rows: fused F(2x2), its own pack kind
        h = ops.conv2d(h, convs[3].weight, convs[3].bias, 1, 1)
        (h.square().mean() + g.square().mean()).backward()
    opt.zero_grad(); fwd_bwd()
    before = packs()
    wino_keys = [k for k in before if isinstance(k[1], tuple) and k[1][0] == 'wino']
    assert len(wino_keys) >= 5                     # forward + data-gradient packs of the wide layers (layer 0 needs no data gradient)
    opt.step()                                      # -> refresh_packs
    after = packs()
    assert set(after) == set(wino_keys), 'only the F(4x4) packs are kept'
    for k in wino_keys:
        assert after[k][1] == before[k][1], 're-packed in place'
        assert after[k][0] == convs[k[0]].weight._version
        assert not torch.equal(after[k][2], before[k][2]), 'the update changed the filters'
        K, C = convs[k[0]].weight.shape[:2]
        ref = torch.empty_like(after[k][2])
        ops.check(ops.lib.fcd_conv_wino_pack(ops._p(convs[k[0]].weight.detach().contiguous()), ops._p(ref), K, C, k[1][1], 4, ops._stream()))
        n_planes_from = 36 * (K if k[1][1] == 0 else C) * (((C if k[1][1] == 0 else K) + 31) // 32 * 32)
        # the single call writes the same region (the bf16 planes behind the fp32 area for these > 64-row layers)
        assert torch.equal(after[k][2][n_planes_from:], ref[n_planes_from:])
    # and the next step runs on the refreshed packs without packing anything again
    opt.zero_grad(); fwd_bwd()
    again = packs()
    for k in wino_keys:
        assert again[k][1] == before[k][1] and torch.equal(again[k][2], after[k][2])


def test_device_hyper_kernels_equal_scalar_kernels():
    """fcd_adam_step_h / fcd_rmsprop_step_h (update-rule scalars read from device memory -- what a caller who captures the step into a hipGraph
    needs, include/fcdgan_hip.h) against the scalar-argument kernels over several steps."""
    import fcd_gan_pytorch_amd as p
    rng = np.random.default_rng(5)
    w0 = torch.from_numpy(rng.standard_normal(10007).astype(np.float32))
    res = {}
    for mode in ('scalar', 'device'):
        for kind in ('adam', 'rmsprop'):
            prm = torch.nn.Parameter(w0.clone().to('cuda'))
            opt = p.optim.Adam([prm], lr=3e-4, betas=(0.9, 0.99)) if kind == 'adam' else p.optim.RMSprop([prm], lr=5e-5)
            if mode == 'device':
                opt.use_device_hyper()
            for it in range(7):
                opt.zero_grad()
                opt.param_groups[0]['lr'] = 3e-4 * (1 + it)
                g = torch.from_numpy(np.random.default_rng(it).standard_normal(10007).astype(np.float32)).to('cuda')
                opt.flat_g.copy_(g)
                opt.step()
            res[mode, kind] = opt.flat_p.cpu().numpy()
    for kind in ('adam', 'rmsprop'):
        np.testing.assert_array_equal(res['scalar', kind], res['device', kind], err_msg=kind)


def test_launch_window_bounds_how_far_the_host_runs_ahead(switches):
    """``_ops._LaunchWindow`` (switch LAUNCH_WINDOW): every 128th C-ABI launch leaves an event, and the issuing thread waits (sleeping) on
    the oldest while more than the window is outstanding -- so at most window / 128 events are ever pending; 0 switches it off."""
    ops = _ops()
    x = rnd(2, 8, 16, 16, seed=5).cuda()
    win = ops._WINDOW
    switches('LAUNCH_WINDOW', 256)
    win.events.clear()                  # (what earlier tests left pending under the default window)
    for _ in range(128 * 6):
        y = ops.bn_act(x, None, ops.ACT_RELU)
        assert len(win.events) <= 2
    assert len(win.events) == 2 and torch.equal(y, torch.relu(x))
    switches('LAUNCH_WINDOW', 0)
    for _ in range(128 * 2):
        ops.bn_act(x, None, ops.ACT_RELU)
    assert len(win.events) == 0
