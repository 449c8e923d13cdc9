"""Drop-in for the reference's ``Module.py`` (same class names, constructor
signatures, ``forward`` signatures and ``state_dict`` keys), running on the
hand-written HIP kernels of ``libfcdgan_hip.so``.

How it differs from a layer-by-layer module tree:
 * parameter holders are stock ``nn.Conv2d`` / ``nn.BatchNorm2d`` / ``nn.PReLU``
   objects (=> identical default initialisation, RNG consumption and checkpoint
   layout) whose arithmetic is never used -- every ``forward`` here calls the
   fused HIP ops directly with the holders' tensors;
 * BatchNorm + ReLU/LeakyReLU/PReLU is one fused op;
 * the Siamese encoder (reference Module.py:114-131) and the Discriminator's
   shared ``net`` (Module.py:220-221) run both branches as ONE batch of 2N
   samples; BN keeps per-branch statistics and ordered running-stat updates via
   ``groups=2`` (see include/fcdgan_hip.h: fcd_bn_act_fwd).
CUDA/ROCm tensors only: there is no CPU fallback in the product.
"""

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

from . import _ops as ops

__all__ = ['DoubleConv', 'Down', 'Up', 'OutConv', 'Segmentor', 'Generator', 'ResidualBlock',
           'Discriminator_SRGAN_simple']


def _conv(holder, x, bn_groups=0):
    """Apply an nn.Conv2d parameter holder through the HIP convolution kernels (``bn_groups``: see ops.conv2d)."""
    return ops.conv2d(x, holder.weight, holder.bias, holder.stride[0], holder.padding[0], bn_groups=bn_groups)


class DoubleConv(nn.Module):
    """(conv3x3 p1 -> BN -> ReLU) x 2 -- reference Module.py:18-35.
    ``double_conv`` keeps the reference's Sequential indices 0,1,(2),3,4,(5)."""

    def __init__(self, in_channels, out_channels, mid_channels=None):
        super().__init__()
        mid = mid_channels if mid_channels else out_channels
        self.double_conv = nn.Sequential(
            nn.Conv2d(in_channels, mid, kernel_size=3, padding=1), nn.BatchNorm2d(mid), nn.ReLU(inplace=True),
            nn.Conv2d(mid, out_channels, kernel_size=3, padding=1), nn.BatchNorm2d(out_channels),
            nn.ReLU(inplace=True))

    def forward(self, x, groups=1, defer=False):
        """``defer`` (train mode only): return the second convolution's output and leave ``double_conv[4]`` + ReLU to the caller, who
        runs them inside the kernels of their consumer (``ops.bn_relu_pool_skip``, ``ops.bn_relu_head``)."""
        s = self.double_conv
        if defer and not self.training:
            raise RuntimeError('DoubleConv.forward(defer=True) is a train-mode path')
        if not self.training and not torch.is_grad_enabled():
            # inference (Demo_RSSS.py:451-491: netS.eval() + no_grad): eval-mode BatchNorm is an
            # affine map per channel -> folded into the conv's filter and bias once, and each
            # conv+BN+ReLU becomes ONE kernel (MFMA conv with the bias+ReLU epilogue)
            (w0, b0), (w1, b1) = self._folded_params()
            return ops.conv2d(ops.conv2d(x, w0, b0, 1, 1, relu=True), w1, b1, 1, 1, relu=True)
        g = groups if self.training else 0        # train-mode BatchNorm behind the conv: statistics out of its output transform
        return self._tail(_conv(s[0], x, bn_groups=g), groups, g, defer=defer)

    def _tail(self, z, groups, g, defer=False):
        """BatchNorm -> ReLU -> conv -> BatchNorm -> ReLU behind the first convolution's output ``z``.  In train mode the first
        BatchNorm + ReLU is applied by the second convolution's loader where that layer runs as F(4x4) (``ops.bn_relu_conv3x3``):
        the activation between the two convolutions is never a tensor."""
        s = self.double_conv
        if ops.bn_relu_conv3x3_ok(z, s[1], s[3].weight, groups):
            z = ops.bn_relu_conv3x3(z, s[1], s[3].weight, s[3].bias, groups=groups, bn_groups=g)
        else:
            z = _conv(s[3], ops.bn_act(z, s[1], ops.ACT_RELU, groups=groups), bn_groups=g)
        if defer:       # the caller applies double_conv[4] + ReLU itself (OutConv.forward_bn: inside the head's kernels)
            return z
        return ops.bn_act(z, s[4], ops.ACT_RELU, groups=groups)

    def forward_pair_cat(self, f2n, up, defer=False):
        """``forward(torch.cat([f2n[:n], f2n[n:], up], dim=1))`` with the first convolution reading the three tensors in
        place (``ops.conv3x3_pair_cat``; caller checked ``ops.conv3x3_pair_cat_ok``)."""
        s = self.double_conv
        if not self.training and not torch.is_grad_enabled():
            (w0, b0), (w1, b1) = self._folded_params()
            return ops.conv2d(ops.conv3x3_pair_cat(f2n, up, w0, b0, relu=True), w1, b1, 1, 1, relu=True)
        g = 1 if self.training else 0
        return self._tail(ops.conv3x3_pair_cat(f2n, up, s[0].weight, s[0].bias, bn_groups=g), 1, g, defer=defer)

    def _folded_params(self):
        s = self.double_conv
        # key: storage + version of every tensor that enters the fold (load_state_dict / in-place
        # edits bump the versions; fcd optimizers only run in train() mode, which drops the cache)
        key = tuple((t.data_ptr(), t._version) for m in (s[0], s[1], s[3], s[4])
                    for t in (m.weight, m.bias) + ((m.running_mean, m.running_var) if hasattr(m, 'running_var') else ()))
        cache = self.__dict__.get('_fcd_folded')
        if cache is not None and self.__dict__.get('_fcd_folded_key') != key:
            cache = None
        if cache is None:
            cache = []
            self.__dict__['_fcd_folded_key'] = key
            with torch.no_grad():
                for conv, bn in ((s[0], s[1]), (s[3], s[4])):
                    scale = bn.weight / torch.sqrt(bn.running_var + bn.eps)
                    w = (conv.weight * scale.view(-1, 1, 1, 1)).contiguous()
                    b = ((conv.bias - bn.running_mean) * scale + bn.bias).contiguous()
                    cache.append((w, b))
            self.__dict__['_fcd_folded'] = cache
        return cache

    def train(self, mode=True):
        self.__dict__.pop('_fcd_folded', None)      # folded filters are only valid for frozen weights
        return super().train(mode)

    def _apply(self, fn, *a, **k):
        self.__dict__.pop('_fcd_folded', None)
        return super()._apply(fn, *a, **k)


class Down(nn.Module):
    """MaxPool2d(2) then DoubleConv -- reference Module.py:38-49."""

    def __init__(self, in_channels, out_channels):
        super().__init__()
        self.maxpool_conv = nn.Sequential(nn.MaxPool2d(2), DoubleConv(in_channels, out_channels))

    def forward(self, x, groups=1):
        return self.maxpool_conv[1](ops.maxpool2(x), groups=groups)

    def forward_skip(self, x, groups=1):
        """``(x, forward(x))`` for an ``x`` that is also a skip connection: use the returned x for the skip (``ops.maxpool2_skip``:
        the two gradients of x are summed inside the max-pool's backward kernel)."""
        skip, pooled = ops.maxpool2_skip(x)
        return skip, self.maxpool_conv[1](pooled, groups=groups)


class Up(nn.Module):
    """x2 upsample (bilinear align_corners=True | ConvTranspose2d k2 s2), zero-pad to
    the skip's size, cat([skip, up]), DoubleConv -- reference Module.py:52-79."""

    def __init__(self, in_channels, out_channels, bilinear=False):
        super().__init__()
        if bilinear:
            self.up = nn.Upsample(scale_factor=2, mode='bilinear', align_corners=True)
            self.conv = DoubleConv(in_channels, out_channels, in_channels // 2)
        else:
            self.up = nn.ConvTranspose2d(in_channels, in_channels // 2, kernel_size=2, stride=2)
            self.conv = DoubleConv(in_channels, out_channels)
        self.bilinear = bool(bilinear)

    def forward(self, x1, x2):
        if self.bilinear:
            x1 = ops.upsample2x(x1)
        else:
            x1 = ops.conv_transpose2x2(x1, self.up.weight, self.up.bias)
        dy, dx = x2.shape[2] - x1.shape[2], x2.shape[3] - x1.shape[3]
        if dy or dx:
            x1 = F.pad(x1, [dx // 2, dx - dx // 2, dy // 2, dy - dy // 2])
        return self.conv(torch.cat([x2, x1], dim=1))

    def forward_pair(self, x1, f2n, defer=False):
        """``forward(x1, cat([f2n[:n], f2n[n:]], dim=1))`` for the Segmentor, whose skip tensor is the channel pair of the two
        temporal branches living in ONE (2N, C, h, w) batch (reference Module.py:116-132): neither concatenation is
        materialised when the first convolution takes tensor lists; otherwise this is ``forward``."""
        n = x1.shape[0]
        if self.bilinear:
            u = ops.upsample2x(x1)
        else:
            u = ops.conv_transpose2x2(x1, self.up.weight, self.up.bias)
        w0 = self.conv.double_conv[0].weight
        if tuple(u.shape[2:]) == tuple(f2n.shape[2:]) and ops.conv3x3_pair_cat_ok(f2n, u, w0):
            return self.conv.forward_pair_cat(f2n, u, defer=defer)
        dy, dx = f2n.shape[2] - u.shape[2], f2n.shape[3] - u.shape[3]
        if dy or dx:
            u = F.pad(u, [dx // 2, dx - dx // 2, dy // 2, dy - dy // 2])
        x = torch.cat([f2n[:n], f2n[n:], u], dim=1)
        if defer:       # (``defer``: train mode only -- the last BatchNorm + ReLU of the DoubleConv is left to the caller)
            c = self.conv
            return c._tail(_conv(c.double_conv[0], x, bn_groups=1), 1, 1, defer=True)
        return self.conv(x)


class OutConv(nn.Module):
    """1x1 conv + sigmoid -> change-density map -- reference Module.py:82-90."""

    def __init__(self, in_channels, out_channels):
        super(OutConv, self).__init__()
        self.conv = nn.Conv2d(in_channels, out_channels, kernel_size=1)
        self.Sigmoid = nn.Sigmoid()

    def forward(self, x):
        if ops.conv1x1_head_supported(x, self.conv.weight):      # one output channel: streaming kernels, sigmoid fused
            return ops.conv1x1_head(x, self.conv.weight, self.conv.bias, sigmoid=True)
        return torch.sigmoid(_conv(self.conv, x))

    def forward_bn(self, z, bn):
        """``forward(relu(bn(z)))`` for a train-mode BatchNorm ``bn`` whose only consumer is this head (the decoder's last DoubleConv,
        reference Module.py:131-133): normalisation, ReLU, 1x1 filter and sigmoid in one pass over z (``ops.bn_relu_head``)."""
        if ops.bn_relu_head_ok(z, bn, self.conv.weight):
            return ops.bn_relu_head(z, bn, self.conv.weight, self.conv.bias, sigmoid=True)
        return self.forward(ops.bn_act(z, bn, ops.ACT_RELU))


class Segmentor(nn.Module):
    """Siamese U-Net -- reference Module.py:93-140."""

    def __init__(self, n_channels, n_outchannels=1, bilinear=False):
        super(Segmentor, self).__init__()
        self.n_channels = n_channels
        self.n_outchannels = n_outchannels
        self.bilinear = bilinear
        factor = 2 if bilinear else 1
        self.inc = DoubleConv(n_channels, 64)
        self.down1 = Down(64, 128)
        self.down2 = Down(128, 256)
        self.down3 = Down(256, 512)
        self.down4 = Down(512, 1024 // factor)
        self.up1 = Up(2048, 1024 // factor, bilinear)
        self.up2 = Up(1024, 512 // factor, bilinear)
        self.up3 = Up(512, 256 // factor, bilinear)
        self.up4 = Up(256, 128, bilinear)
        self.outc = OutConv(128, n_outchannels)

    @staticmethod
    def _pair(f, n):
        # (2N,C,h,w) two-branch features -> (N,2C,h,w) = cat([branch1, branch2], dim=1)
        # (split, not two slices: its backward is ONE concatenation of the branch gradients instead of two zero-filled full-size
        #  tensors and an add)
        a, b = f.split(n, dim=0)
        return torch.cat([a, b], dim=1)

    def forward(self, x1, x2):
        n = x1.shape[0]
        with ops.batched_bn_counters():                              # the 18 num_batches_tracked increments as one launch
            f, pending = self._double_conv(self.inc, torch.cat([x1, x2], dim=0))      # both temporal branches in one batch
            return self._after_inc(f, n, pending)

    @torch.no_grad()
    def forward_raw(self, x1_raw, x2_raw, valid, stats):
        """Inference on RAW (un-normalised) tiles: the per-band ``(x - mean) / std`` of NORMALIZE (CommonFunc.py:199-224)
        is folded into the first convolution instead of being a pass of its own.  ``stats`` = (meanX, stdX, meanY,
        stdY); ``valid`` (N,1,H,W) = 1 where the patch holds scene pixels (the reference embeds the NORMALISED block
        in a zero patch, data_utils.py:106-116, so padding is 0 in normalised space, i.e. ``mean`` in raw space).

        Exact at the borders: the input gets one extra channel holding ``valid`` whose filter taps are
        ``-sum_c w[k][c][r][s] * mean_c / std_c``, the band filters are divided by ``std_c``.  The convolution's own
        zero padding zeroes that channel too, so every tap contributes ``w * (x - mean) / std`` inside the scene and
        0 outside -- the same sum as the reference's, up to fp32 rounding of the re-associated products.
        eval() + folded BatchNorm only (the Demo_RSSS / Demo_USSS inference path)."""
        if self.training:
            raise RuntimeError('Segmentor.forward_raw is the eval-mode inference path (call .eval() first)')
        (w0, b0), (w1, b1) = self.inc._folded_params()
        n = x1_raw.shape[0]
        feats = []
        # the augmented first-layer filters depend on the folded weights and the statistics only: built once per (weights,
        # statistics) and kept on the module, so the packed-filter cache of ops.conv2d (keyed on the tensor object) hits on
        # every later batch instead of re-folding and re-packing per call
        # (the entry HOLDS w0: a folded tensor rebuilt after a train()/eval() toggle may reuse the freed one's address, so
        # identity is tested on the live object, never on id(); the fold's own key -- storage + version of every source
        # parameter and BatchNorm buffer -- rides along)
        key = (self.inc.__dict__.get('_fcd_folded_key'), w0._version,
               tuple(tuple(float(v) for v in np.asarray(t, dtype=np.float64).reshape(-1)) for t in stats))
        hit = self.__dict__.get('_fcd_raw_filters')
        if hit is None or hit[0] != key or hit[2] is not w0:
            augs = []
            for mean, std in ((stats[0], stats[1]), (stats[2], stats[3])):
                m = torch.as_tensor(mean, dtype=torch.float64, device=w0.device)[:w0.shape[1]]
                s = torch.as_tensor(std, dtype=torch.float64, device=w0.device)[:w0.shape[1]]
                wd = w0.double()
                augs.append(torch.cat([wd / s.view(1, -1, 1, 1), -(wd * (m / s).view(1, -1, 1, 1)).sum(dim=1, keepdim=True)],
                                      dim=1).float().contiguous())
            hit = (key, augs, w0)
            self.__dict__['_fcd_raw_filters'] = hit
        for x, w_aug in ((x1_raw, hit[1][0]), (x2_raw, hit[1][1])):
            xa = torch.cat([x, valid.to(x.dtype)], dim=1)
            feats.append(ops.conv2d(xa, w_aug, b0, 1, 1, relu=True))
        f = ops.conv2d(torch.cat(feats, dim=0), w1, b1, 1, 1, relu=True)
        return self._after_inc(f, n)

    def train(self, mode=True):
        self.__dict__.pop('_fcd_raw_filters', None)      # built from the folded (frozen-weight) filters: dropped with them
        return super().train(mode)

    def _apply(self, fn, *a, **k):
        self.__dict__.pop('_fcd_raw_filters', None)
        return super()._apply(fn, *a, **k)

    @staticmethod
    def _double_conv(dc, x):
        """An encoder DoubleConv on the two-branch batch: (tensor, the BatchNorm still to be applied to it or None)."""
        if dc.training and torch.is_grad_enabled():
            return dc(x, groups=2, defer=True), dc.double_conv[4]
        return dc(x, groups=2), None

    def _after_inc(self, f, n, pending=None):
        # f: both temporal branches in one (2N, C, h, w) batch; the reference's skip tensor is cat([branch1, branch2], 1)
        # (Module.py:116-132) -- kept as the un-paired batch and read in place by the decoder's first convolutions
        # ``pending``: the BatchNorm (+ ReLU) that has not been applied to f yet -- in train mode the last BatchNorm of a DoubleConv
        # runs inside the kernels of what consumes it: with the max-pool and the skip tensor here (ops.bn_relu_pool_skip: one node
        # that also sums the level's two gradients), inside the head at the end
        feats = []
        for stage in (self.down1, self.down2, self.down3, self.down4):
            if pending is not None and ops.bn_relu_pool_skip_ok(f, pending, 2):
                skip, pooled = ops.bn_relu_pool_skip(f, pending, groups=2)
            else:
                if pending is not None:
                    f = ops.bn_act(f, pending, ops.ACT_RELU, groups=2)
                skip, pooled = ops.maxpool2_skip(f)      # skip IS f, as the output of the node that sums f's two gradients
            feats.append(skip)
            f, pending = self._double_conv(stage.maxpool_conv[1], pooled)
        if pending is not None:
            f = ops.bn_act(f, pending, ops.ACT_RELU, groups=2)
        feats.append(f)
        x = self.up1.forward_pair(self._pair(feats[4], n), feats[3])
        x = self.up2.forward_pair(x, feats[2])
        x = self.up3.forward_pair(x, feats[1])
        if self.training and self.up4.conv.training:       # the last BatchNorm + ReLU runs inside the head's kernels
            return self.outc.forward_bn(self.up4.forward_pair(x, feats[0], defer=True), self.up4.conv.double_conv[4])
        x = self.up4.forward_pair(x, feats[0])
        return self.outc(x)


class ResidualBlock(nn.Module):
    """conv-BN-PReLU-conv-BN + identity -- reference Module.py:174-190."""

    def __init__(self, channels):
        super(ResidualBlock, self).__init__()
        self.conv1 = nn.Conv2d(channels, channels, kernel_size=3, padding=1)
        self.bn1 = nn.BatchNorm2d(channels)
        self.prelu = nn.PReLU()
        self.conv2 = nn.Conv2d(channels, channels, kernel_size=3, padding=1)
        self.bn2 = nn.BatchNorm2d(channels)

    def forward(self, x):
        r = ops.bn_act(_conv(self.conv1, x), self.bn1, ops.ACT_PRELU, slope=self.prelu.weight)
        r = ops.bn_act(_conv(self.conv2, r), self.bn2, ops.ACT_NONE)
        return x + r


class Generator(nn.Module):
    """SRGAN-style generator, raw (no tanh) output -- reference Module.py:142-172."""

    def __init__(self, n_channels):
        super(Generator, self).__init__()
        self.block1 = nn.Sequential(nn.Conv2d(n_channels, 64, kernel_size=9, padding=4), nn.PReLU())
        self.block2 = ResidualBlock(64)
        self.block3 = ResidualBlock(64)
        self.block4 = ResidualBlock(64)
        self.block5 = ResidualBlock(64)
        self.block6 = ResidualBlock(64)
        self.block7 = nn.Sequential(nn.Conv2d(64, 64, kernel_size=3, padding=1), nn.BatchNorm2d(64))
        self.block8 = nn.Conv2d(64, n_channels, kernel_size=9, padding=4)

    def _folded(self):
        """eval-mode BN folded into the preceding conv (filters cached, keyed by tensor versions)."""
        pairs = [(blk.conv1, blk.bn1) for blk in (self.block2, self.block3, self.block4, self.block5, self.block6)]
        pairs += [(blk.conv2, blk.bn2) for blk in (self.block2, self.block3, self.block4, self.block5, self.block6)]
        pairs.append((self.block7[0], self.block7[1]))
        key = tuple((t.data_ptr(), t._version) for c, b in pairs
                    for t in (c.weight, c.bias, b.weight, b.bias, b.running_mean, b.running_var))
        hit = self.__dict__.get('_fcd_folded')
        if hit is None or hit[0] != key:
            out = []
            with torch.no_grad():
                for conv, bn in pairs:
                    scale = bn.weight / torch.sqrt(bn.running_var + bn.eps)
                    out.append(((conv.weight * scale.view(-1, 1, 1, 1)).contiguous(),
                                ((conv.bias - bn.running_mean) * scale + bn.bias).contiguous()))
            hit = (key, out)
            self.__dict__['_fcd_folded'] = hit
        return hit[1]

    def _infer(self, x):
        """eval + no_grad (the adversarial phases, Demo_RSSS.py:240,319 / Demo_WSSS.py:206,307): 13
        kernels, no elementwise pass -- BN folded, PReLU and the skip adds in the conv epilogues."""
        f = self._folded()
        c1 = self.block1[0]
        b1 = ops.conv2d_infer(x, c1.weight, c1.bias, 1, 4, ops.ACT_PRELU, slope=self.block1[1].weight)
        h = b1
        for i, blk in enumerate((self.block2, self.block3, self.block4, self.block5, self.block6)):
            r = ops.conv2d_infer(h, f[i][0], f[i][1], 1, 1, ops.ACT_PRELU, slope=blk.prelu.weight)
            h = ops.conv2d_infer(r, f[5 + i][0], f[5 + i][1], 1, 1, ops.ACT_NONE, residual=h)
        s = ops.conv2d_infer(h, f[10][0], f[10][1], 1, 1, ops.ACT_NONE, residual=b1)
        return ops.conv2d_infer(s, self.block8.weight, self.block8.bias, 1, 4)

    def train(self, mode=True):
        self.__dict__.pop('_fcd_folded', None)
        return super().train(mode)

    def _apply(self, fn, *a, **k):
        self.__dict__.pop('_fcd_folded', None)
        return super()._apply(fn, *a, **k)

    def forward(self, x):
        if not self.training and not torch.is_grad_enabled():
            return self._infer(x)
        with ops.batched_bn_counters():
            b1 = ops.bn_act(_conv(self.block1[0], x), None, ops.ACT_PRELU, slope=self.block1[1].weight)
            h = b1
            for blk in (self.block2, self.block3, self.block4, self.block5, self.block6):
                h = blk(h)
            h = ops.bn_act(_conv(self.block7[0], h), self.block7[1], ops.ACT_NONE)
            return _conv(self.block8, b1 + h)


class Discriminator_SRGAN_simple(nn.Module):
    """4x stride-2 conv3x3 (+BN) + LeakyReLU(0.2) shared by both inputs, classifier on
    the feature DIFFERENCE -- reference Module.py:192-223."""

    def __init__(self, n_channels=3):
        super(Discriminator_SRGAN_simple, self).__init__()
        self.net = nn.Sequential(
            nn.Conv2d(n_channels, 64, kernel_size=3, stride=2, padding=1), nn.LeakyReLU(0.2, inplace=True),
            nn.Conv2d(64, 128, kernel_size=3, stride=2, padding=1), nn.BatchNorm2d(128),
            nn.LeakyReLU(0.2, inplace=True),
            nn.Conv2d(128, 256, kernel_size=3, stride=2, padding=1), nn.BatchNorm2d(256),
            nn.LeakyReLU(0.2, inplace=True),
            nn.Conv2d(256, 512, kernel_size=3, stride=2, padding=1), nn.BatchNorm2d(512),
            nn.LeakyReLU(0.2, inplace=True))
        self.classifier = nn.Sequential(
            nn.AdaptiveAvgPool2d(1), nn.Conv2d(512, 1024, kernel_size=1), nn.LeakyReLU(0.2, inplace=True),
            nn.Conv2d(1024, 1, kernel_size=1))

    def features(self, z, groups, order=None):
        """Shared ``net`` on a batch made of ``groups`` independent calls (BN statistics
        per call, running stats updated call by call -- in the sequence ``order`` when given)."""
        s = self.net
        z = ops.bn_act(_conv(s[0], z), None, ops.ACT_LEAKY, slope_imm=0.2, groups=groups)
        for ci, bi in ((2, 3), (5, 6), (8, 9)):
            z = ops.bn_act(_conv(s[ci], z), s[bi], ops.ACT_LEAKY, slope_imm=0.2, groups=groups, order=order)
        return z

    def classify(self, diff):
        """Classifier on a feature difference (N, 512, h, w) -- or on its global average (N, 512, 1, 1)."""
        c = self.classifier
        d = diff if diff.shape[2:] == (1, 1) else diff.mean(dim=(2, 3), keepdim=True)
        d = ops.bn_act(_conv(c[1], d), None, ops.ACT_LEAKY, slope_imm=0.2)
        return torch.sigmoid(_conv(c[3], d).view(diff.shape[0]))

    # How the pooled pair difference is formed (A/B switch for tools/parity_probe_d.py; the product runs 'fused'):
    #  'fused'  -- ops.pair_gap_diff: element-wise f_x - f_y first (the reference's order, Module.py:222-223), averaged with an
    #              fp64 accumulator in one kernel, classifier on all pairs as one batch;
    #  'diff'   -- the same order on ATen ops in fp32: (f_x - f_y).mean();
    #  'pooled' -- round 3: mean(f) of the whole batch first, then the difference of two ROUNDED means of nearly equal
    #              features (loses the bits the element-wise difference keeps).
    POOL_MODE = None          # None: follow the switch D_POOL (0 fused / 1 diff / 2 pooled); a string pins it for this class / instance

    def _classify_pairs(self, f, npairs):
        n = f.shape[0] // (2 * npairs)
        mode = self.POOL_MODE or ('fused', 'diff', 'pooled')[ops.switch('D_POOL')]
        if mode == 'fused':
            out = self.classify(ops.pair_gap_diff(f, npairs))
            return list(out.split(n, dim=0)) if npairs > 1 else [out]
        if mode == 'pooled':
            f = f.mean(dim=(2, 3), keepdim=True)
        return [self.classify(f[(2 * i) * n:(2 * i + 1) * n] - f[(2 * i + 1) * n:(2 * i + 2) * n]) for i in range(npairs)]

    def forward(self, x, y):
        with ops.batched_bn_counters():
            return self._classify_pairs(self.features(torch.cat([x, y], dim=0), groups=2), 1)[0]

    def forward_pairs(self, pairs):
        """Evaluate several (x, y) pairs in one batched pass; equivalent to calling
        ``forward`` on each pair in order (BN running stats see x1,y1,x2,y2,...)."""
        return self.forward_stacked(torch.cat([t for p in pairs for t in p], dim=0), len(pairs))

    def forward_shared_first(self, z, nsecond):
        """``[forward(x, y_1), ..., forward(x, y_k)]`` for ``z`` = cat([x, y_1, ..., y_k], dim=0): the reference's k calls with the SAME
        first argument (Demo_RSSS.py:293,302: ``netD(x_mask, y_mask)``, ``netD(x_unc, y_unc)`` with x_unc == x_mask) run ``net`` on x
        k times in train mode -- the same samples, weights and batch statistics, hence the same features.  Here x goes through
        ``net`` once: its features feed every pair (autograd sums the k gradients before the one backward pass through ``net``,
        which is what the k separate passes add up to), and the BatchNorm running statistics replay the reference's call order
        x, y_1, x, y_2, ...  Falls back to ``forward_stacked`` on the repeated batch where that replay is not available (eval
        mode, SyncBN)."""
        k = int(nsecond)
        if z.shape[0] % (k + 1):
            raise ValueError('forward_shared_first: %d samples are not 1 + %d equal groups' % (z.shape[0], k))
        n = z.shape[0] // (k + 1)
        # (the kernel's replay order is sixteen 4-bit entries -- include/fcdgan_hip.h, fcd_bn_act_fwd_replay: more than 8 second
        #  arguments take the repeated batch as well)
        if not self.training or ops.sync_bn_active() or k + 1 > 16 or 2 * k > 16:
            rep = torch.cat([t for i in range(k) for t in (z[:n], z[(i + 1) * n:(i + 2) * n])], dim=0)
            return self.forward_stacked(rep, k)
        order = [g for i in range(k) for g in (0, i + 1)]
        with ops.batched_bn_counters():
            f = self.features(z, groups=k + 1, order=order)
            parts = f.split(n, dim=0)       # backward: one cat of the group gradients (x's two uses summed first), no zero-filled slices
            pairs = torch.cat([t for i in range(k) for t in (parts[0], parts[i + 1])], dim=0)
            return self._classify_pairs(pairs, k)

    def forward_stacked(self, z, npairs):
        """``forward_pairs`` on the already batched tensor ``z`` = cat([x1, y1, x2, y2, ...], dim=0) (what
        ``ops.masked_stack`` writes): a list of ``npairs`` outputs."""
        if z.shape[0] % (2 * npairs):
            raise ValueError('forward_stacked: %d samples are not %d (x, y) pairs' % (z.shape[0], npairs))
        with ops.batched_bn_counters():
            return self._classify_pairs(self.features(z, groups=2 * npairs), npairs)
