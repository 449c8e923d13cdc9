"""ReLU / max-pool DECISIONS of one conv plan injected into a forward + backward pass that runs on another.

``InjectedConvRelu`` computes a frozen conv + ReLU (+ MaxPool2d(2)) layer with the VALUES of the plan in force and saves the
ReLU mask / pool argmax code of the DIRECT kernels' forward pass for the backward; ``inject_direct_vgg_decisions()`` routes every
frozen 3x3 layer of the perception VGG through it.  Used by tests/test_gpu_fullsize_bwd.py and tools/parity_probe_g.py."""
import contextlib

import torch


class InjectedConvRelu(torch.autograd.Function):
    """relu(conv3x3(x) + b) [+ MaxPool2d(2)] of a FROZEN layer with the VALUES of the plan in force and the DECISIONS (ReLU mask /
    pool argmax code) of the direct kernels' forward pass: the backward gates / routes with the direct plan's decisions."""

    @staticmethod
    def forward(ctx, x, weight, bias, pool):
        from fcd_gan_pytorch_amd import _ops as ops
        lib = ops.lib
        d = ops._desc(x.shape, weight.shape, 1, 1)
        dev = x.device

        def run():
            if pool:
                yp = torch.empty((d.N, d.K, d.P // 2, d.Q // 2), device=dev)
                code = torch.empty(yp.shape, dtype=torch.uint8, device=dev)
                ops._fwd_conv(d, x, weight, bias, None, True, pool_y=yp, code=code)
                return yp, code
            y = torch.empty((d.N, d.K, d.P, d.Q), device=dev)
            if d.C <= 4:                                   # the thin first layer has one plan only
                ops.check(lib.fcd_conv2d_fwd(ops.ctypes.byref(d), ops._p(x), ops._p(ops.packed_weight(weight, 0)), ops._p(bias), ops._p(y), 1,
                                             ops._stream()), 'fcd_conv2d_fwd')
            else:
                ops._fwd_conv(d, x, weight, bias, y, True)
            return y, None
        plan = lib.fcd_conv_wino_set(-1)
        val, _ = run()
        lib.fcd_conv_wino_set(0)
        ref, code = run()
        lib.fcd_conv_wino_set(plan)
        ctx.save_for_backward(weight, code if pool else ref)
        ctx.geom = (tuple(x.shape), bool(pool))
        return val

    @staticmethod
    def backward(ctx, dy):
        from fcd_gan_pytorch_amd import _ops as ops
        weight, dec = ctx.saved_tensors
        xshape, pool = ctx.geom
        d = ops._desc(xshape, weight.shape, 1, 1)
        dx = torch.empty(xshape, device=dy.device)
        dy = dy.contiguous()
        if pool:
            ops._bwd_data_conv(d, dy, weight, dx, code=dec)
        else:
            ops._bwd_data_conv(d, dy, weight, dx, yrelu=dec)
        return dx, None, None, None


@contextlib.contextmanager
def inject_direct_vgg_decisions():
    from fcd_gan_pytorch_amd import _ops as ops
    orig_conv, orig_pool, orig_chain = ops.conv2d, ops.conv2d_relu_maxpool2, ops.frozen_chain_ok
    ops.conv2d = lambda x_, w_, b_=None, stride=1, padding=0, relu=False, bn_groups=0: (
        InjectedConvRelu.apply(x_, w_, b_, False) if (relu and not w_.requires_grad and tuple(w_.shape[2:]) == (3, 3))
        else orig_conv(x_, w_, b_, stride, padding, relu=relu, bn_groups=bn_groups))
    ops.conv2d_relu_maxpool2 = lambda x_, w_, b_: (InjectedConvRelu.apply(x_, w_, b_, True) if not w_.requires_grad else orig_pool(x_, w_, b_))
    ops.frozen_chain_ok = lambda *a, **k: False          # layer by layer: every layer is its own node
    try:
        yield
    finally:
        ops.conv2d, ops.conv2d_relu_maxpool2, ops.frozen_chain_ok = orig_conv, orig_pool, orig_chain
