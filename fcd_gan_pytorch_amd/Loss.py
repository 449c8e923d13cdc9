"""Drop-in for the reference's ``Loss.py``: ``PerceptionLoss``, ``CNetLoss``,
``CGeneratorLoss``, ``region_loss`` with the same constructor / forward
signatures and return-tuple orders, on the HIP kernels.

 * the masked reconstruction terms (reference Loss.py:76-84 L1, :110-119 MSE) and
   region_loss (:127-141) are ONE fused per-sample reduction kernel each
   (csrc/losses.hip) -- no Python loop over samples, no ``num_wnc[i] == 0`` host
   sync; the skip-if-zero / divide logic runs on the N per-sample numbers on
   device;
 * MS-SSIM is the fused level kernel of ``ssim.py``;
 * the VGG16 perception term runs every band (and both images) as ONE batch
   through the MFMA conv kernels instead of a Python loop over bands.
"""
import os
import warnings

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import _lib
from . import _ops as ops
from .ssim import MS_SSIM

_VGG_CFG = (64, 64, 'M', 128, 128, 'M', 256, 256, 256, 'M', 512, 512, 512, 'M', 512, 512, 512, 'M')
_TAPS = (29, 22, 15, 8, 3)     # reference Loss.py:30


def _vgg16_features():
    """torchvision's vgg16().features layout (31 entries: conv/ReLU/MaxPool) as
    parameter holders, so ``state_dict`` keys match the reference's ``self.net``."""
    layers, cin = [], 3
    for v in _VGG_CFG:
        if v == 'M':
            layers.append(nn.MaxPool2d(kernel_size=2, stride=2))
        else:
            layers += [nn.Conv2d(cin, v, kernel_size=3, padding=1), nn.ReLU(inplace=True)]
            cin = v
    return nn.Sequential(*layers)


_HUB_FILE = 'vgg16-397923af.pth'      # torchvision's ImageNet checkpoint (the file vgg16(pretrained=True) caches)


def _vgg_checkpoint_path():
    """Where the ImageNet VGG16 weights can come from offline, in order: the file named by
    FCDGAN_VGG16_WEIGHTS (a missing file is an error, not a fallback), then torchvision's hub cache
    ($TORCH_HOME/hub/checkpoints, ~/.cache/torch/hub/checkpoints)."""
    path = os.environ.get('FCDGAN_VGG16_WEIGHTS', '')
    if path:
        if not os.path.exists(path):
            raise FileNotFoundError('FCDGAN_VGG16_WEIGHTS=%s does not exist' % path)
        return path
    roots = [os.path.join(os.environ['TORCH_HOME'], 'hub')] if os.environ.get('TORCH_HOME') else []
    roots.append(os.path.join(os.path.expanduser('~'), '.cache', 'torch', 'hub'))
    for r in roots:
        cand = os.path.join(r, 'checkpoints', _HUB_FILE)
        if os.path.exists(cand):
            return cand
    return None


def _load_vgg_weights(net, allow_seeded):
    """The reference downloads torchvision's ImageNet checkpoint (Loss.py:25).  There is no network
    here: the weights come from FCDGAN_VGG16_WEIGHTS (a torch-saved state_dict of ``vgg16()`` or of
    ``vgg16().features``: keys 'features.0.weight'... or '0.weight'...) or from torchvision's hub cache.
    Without either the constructor RAISES -- a perception term on random filters is a different loss,
    not a degraded one -- unless the caller passes ``allow_seeded=True`` (tests, benchmarks), which gives
    the stack a fixed-seed He initialisation.  Returns True when ImageNet weights were loaded."""
    path = _vgg_checkpoint_path()
    if path:
        sd = torch.load(path, map_location='cpu')
        sd = {k.replace('features.', ''): v for k, v in sd.items() if 'classifier' not in k}
        net.load_state_dict(sd)
        return True
    if not allow_seeded:
        raise RuntimeError(
            'PerceptionLoss: the ImageNet VGG16 weights the reference uses (Loss.py:25, vgg16(pretrained=True)) are '
            'not available offline.  Set FCDGAN_VGG16_WEIGHTS to a torch-saved state_dict of torchvision\'s vgg16 '
            '(or place %s in the torch hub cache), or pass allow_seeded=True to run on fixed-seed random filters '
            '(parity tests / benchmarks only: the loss value then differs from the reference\'s).' % _HUB_FILE)
    warnings.warn('PerceptionLoss: allow_seeded=True -- VGG16 runs on a fixed-seed He initialisation, NOT on the '
                  'ImageNet weights of the reference.')
    g = torch.Generator().manual_seed(16)
    with torch.no_grad():
        for m in net:
            if isinstance(m, nn.Conv2d):
                fan_in = m.weight.shape[1] * 9
                m.weight.copy_(torch.randn(m.weight.shape, generator=g) * (2.0 / fan_in) ** 0.5)
                m.bias.zero_()
    return False


class PerceptionLoss(nn.Module):
    """VGG16 feature MSE -- reference Loss.py:17-61.  ``allow_seeded`` (not in the reference): see
    :func:`_load_vgg_weights`; ``self.pretrained`` tells which weights the stack carries."""

    def __init__(self, feature_layer=1, perception_perBand=False, allow_seeded=False):
        super(PerceptionLoss, self).__init__()
        vgg = _vgg16_features().eval()
        self.pretrained = _load_vgg_weights(vgg, allow_seeded)
        for param in vgg.parameters():
            param.requires_grad = False
        self.net = vgg
        feature_layer = feature_layer if feature_layer > 0 else 1
        feature_layer = feature_layer if feature_layer < 6 else 5
        self.feature_layer_list = list(_TAPS[:feature_layer])
        self.perception_perBand = perception_perBand
        self.loss = nn.MSELoss()

    def _first_filter_1ch(self):
        """A band replicated to 3 channels (Loss.py:52-53) sees the first conv as ONE-channel
        conv with the filter summed over its input channels: same map, a third of the reads,
        and its data gradient is directly the gradient of the band."""
        w = self.net[0].weight
        hit = self.__dict__.get('_fcd_w1')
        if hit is None or hit[0] != w._version or hit[1].device != w.device:
            hit = (w._version, w.detach().sum(dim=1, keepdim=True).contiguous())
            self.__dict__['_fcd_w1'] = hit
        return hit[1]

    def _features(self, z, single_band=False):
        """All 31 layers (the reference runs the stack to the end, Loss.py:45-49);
        returns {tap index: activation}."""
        taps = {}
        layers = list(self.net)
        i = 0
        while i < len(layers):
            layer = layers[i]
            run = self._frozen_run(layers, i) if isinstance(layer, nn.Conv2d) and i > 0 else None
            if run is not None and ops.frozen_chain_ok(z, [c.weight for c in run[0]]):
                # conv + ReLU pairs up to the next max-pool / tapped activation as ONE node: the activations between them
                # are never written (ops.frozen_conv_chain)
                convs, pool, nxt = run
                z = ops.frozen_conv_chain(z, [c.weight for c in convs], [c.bias for c in convs], pool=pool)
                last = nxt - 1              # index of the entry that produced z (ReLU, or the fused max-pool)
                if last in self.feature_layer_list:
                    taps[last] = z
                i = nxt
                continue
            if isinstance(layer, nn.Conv2d):
                fuse = i + 1 < len(layers) and isinstance(layers[i + 1], nn.ReLU)
                # conv + bias + ReLU in one kernel (epilogue); the ReLU mask is re-derived from
                # the output inside the data-gradient kernel's loader
                wgt = self._first_filter_1ch() if (i == 0 and single_band) else layer.weight
                pool_next = fuse and i + 2 < len(layers) and isinstance(layers[i + 2], nn.MaxPool2d)
                if pool_next and (i + 1) not in self.feature_layer_list:
                    # conv + bias + ReLU + MaxPool in ONE kernel: the full-resolution activation is
                    # never written, its backward needs one code byte per pooled element
                    z = ops.conv2d_relu_maxpool2(z, wgt, layer.bias)
                    i += 3
                    continue
                z = ops.conv2d(z, wgt, layer.bias, 1, 1, relu=fuse)
                if fuse:
                    i += 1          # the ReLU entry (tapped indices are all ReLU outputs)
            elif isinstance(layer, nn.ReLU):
                z = ops.bn_act(z, None, ops.ACT_RELU)
            else:
                z = ops.maxpool2(z)
            if i in self.feature_layer_list:
                taps[i] = z
            i += 1
        return taps

    def _frozen_run(self, layers, i):
        """The run of (Conv2d, ReLU) pairs starting at entry ``i`` with no tapped activation inside: (convs, pool, next index),
        ``pool`` = the run ends in a MaxPool2d that is fused into its last layer (only when that layer's ReLU is not tapped);
        None for runs shorter than two layers or frozen-ness / bias conditions the chain does not cover."""
        convs, k = [], i
        while (k + 1 < len(layers) and isinstance(layers[k], nn.Conv2d) and isinstance(layers[k + 1], nn.ReLU)
               and not layers[k].weight.requires_grad and layers[k].bias is not None and not layers[k].bias.requires_grad):
            convs.append(layers[k])
            k += 2
            if (k - 1) in self.feature_layer_list:
                break
        if len(convs) < 2:
            return None
        pool = (k < len(layers) and isinstance(layers[k], nn.MaxPool2d) and (k - 1) not in self.feature_layer_list
                and k not in self.feature_layer_list)
        return convs, pool, (k + 1 if pool else k)

    def forward(self, target_image, generate_image, cmask, stacked=None):
        """``stacked``: ``masked_pair(target_image, generate_image, cmask)`` when the caller has it already (the criteria
        feed the same masked pair to MS-SSIM)."""
        n = target_image.shape[0]
        nl = len(self.feature_layer_list)
        if not self.perception_perBand:
            assert target_image.shape[1] >= 3
            if target_image.shape[1] == 3:
                z = stacked if stacked is not None else masked_pair(target_image, generate_image, cmask)
            else:
                keep = 1 - cmask
                z = torch.cat([target_image[:, 0:3] * keep, generate_image[:, 0:3] * keep], dim=0)
            nb = n
        else:
            # every band becomes its own 3-channel image (band replicated), all bands of
            # both images in one batch: rows [0, n*C) target, [n*C, 2*n*C) generated
            C, H, W = target_image.shape[1:]
            z = stacked if stacked is not None else masked_pair(target_image, generate_image, cmask)
            z = z.reshape(2 * n * C, 1, H, W)           # see _first_filter_1ch
            nb = n * C
        feats = self._features(z, single_band=self.perception_perBand)
        total = 0
        for i in sorted(self.feature_layer_list):
            f = feats[i]
            # sum over bands of per-band MSE / C  ==  MSE over the band-batched tensor
            total = total + mse_halves(f, nb) / nl
        return total


def masked_pair(target_image, generate_image, cmask):
    """``torch.cat([target_image * (1 - cmask), generate_image * (1 - cmask)], dim=0)`` -- the reference's
    ``mask_t = target * (1 - cmask).repeat(...)`` / ``mask_g`` (Loss.py:78-79,111-112) as one batch, from one HIP kernel
    (``ops.masked_stack``; switch FUSED_GLUE=0: the ATen sequence)."""
    if not _lib.switch('FUSED_GLUE') or not target_image.is_cuda:
        keep = 1 - cmask
        return torch.cat([target_image * keep, generate_image * keep], dim=0)
    return ops.masked_stack([target_image, generate_image], cmask)


class _MseHalves(torch.autograd.Function):
    """``F.mse_loss(f[:nb], f[nb:])`` of a band-batched feature tensor as ONE node.  Through autograd the two slices cost a
    ``slice_backward`` each (a zero-filled full-size tensor plus a copy) and an add of the two: ten launches and five passes over
    the 208-image tap per step; here the backward writes both halves of ``grad f`` directly (reference Loss.py:57-59 computes
    the same mean over separate tensors)."""

    @staticmethod
    def forward(ctx, f, nb):
        d = f[:nb] - f[nb:]
        ctx.save_for_backward(d)
        ctx.n_total = f.shape[0]
        flat = d.reshape(-1)
        return torch.dot(flat, flat) / flat.numel()

    @staticmethod
    def backward(ctx, gout):
        d, = ctx.saved_tensors
        nb = d.shape[0]
        c = gout * (2.0 / d.numel())
        g = torch.empty((ctx.n_total,) + tuple(d.shape[1:]), dtype=d.dtype, device=d.device)
        torch.mul(d, c, out=g[:nb])
        torch.mul(d, -c, out=g[nb:])
        return g, None


def mse_halves(f, nb):
    """mean((f[:nb] - f[nb:]) ** 2); ``f`` holds exactly ``2 * nb`` rows."""
    if f.shape[0] != 2 * nb:
        raise ValueError('mse_halves: %d rows for two halves of %d' % (f.shape[0], nb))
    if not _lib.switch('FUSED_GLUE') or not f.is_cuda:
        return F.mse_loss(f[:nb], f[nb:])
    return _MseHalves.apply(f, nb)


def _masked_ratio(a, b, m, kind, complement, scale, skip_zero):
    """mean_i( num_i * scale / wsum_i ) of the masked sums -- one autograd node (``ops.masked_ratio_mean``); switch FUSED_GLUE=0: the
    two halves through ATen (``ops.masked_sums`` + :func:`_per_sample_ratio`)."""
    if not _lib.switch('FUSED_GLUE'):
        num, wsum = ops.masked_sums(a, b, m, kind, complement)
        return _per_sample_ratio(num, wsum, scale, skip_zero)
    return ops.masked_ratio_mean(a, b, m, kind, complement, scale, skip_zero)


def _per_sample_ratio(num, wsum, scale, skip_zero):
    """mean_i( num_i * scale / wsum_i ), optionally skipping wsum_i == 0 samples
    (reference Loss.py:115-119 `continue`) -- on device, no host sync."""
    if skip_zero:
        ok = wsum != 0
        terms = torch.where(ok, num * scale / torch.where(ok, wsum, torch.ones_like(wsum)), torch.zeros_like(num))
    else:
        terms = num * scale / wsum
    return terms.sum() / num.shape[0]


class CNetLoss(nn.Module):
    """USSS criterion -- reference Loss.py:64-95.  Returns
    (generator_loss, l1_loss, perception_loss, ssim_loss)."""

    def __init__(self, channel=4, perception_layer=1, perception_perBand=True, allow_seeded=False):
        super(CNetLoss, self).__init__()
        self.mse = nn.MSELoss()
        self.loss_generator = nn.L1Loss()
        self.loss_perception = PerceptionLoss(feature_layer=perception_layer, perception_perBand=perception_perBand,
                                              allow_seeded=allow_seeded)
        self.ssim = MS_SSIM(data_range=1.0, channel=channel)

    def forward(self, target_image, generate_image, cmap, generator_mask_switch=False):
        C = target_image.shape[1]
        cmask = (torch.sign(cmap - 0.5) + 1) / 2
        # L1(mask_t_i, mask_g_i) * HW / num_wnc_i  ==  num_i / (C * wsum_i)   (no zero guard, as the reference)
        generator_loss = _masked_ratio(target_image, generate_image, cmap, 0, True, 1.0 / C, skip_zero=False)
        l1_loss = torch.mean(abs(cmap))
        n = target_image.shape[0]
        z = masked_pair(target_image, generate_image, cmap)
        perception_loss = self.loss_perception(target_image, generate_image, cmask if generator_mask_switch else cmap,
                                               stacked=None if generator_mask_switch else z)
        ssim_loss = 1 - self.ssim(*z.split(n, dim=0))
        return generator_loss, l1_loss, perception_loss, ssim_loss


class CGeneratorLoss(nn.Module):
    """WSSS / RSSS criterion -- reference Loss.py:100-124.  Returns
    (generator_loss, ssim_loss, perception_loss)."""

    def __init__(self, channel=3, perception_layer=1, perception_perBand=False, allow_seeded=False):
        super(CGeneratorLoss, self).__init__()
        self.loss_generator = nn.MSELoss()
        self.ssim = MS_SSIM(data_range=1.0, channel=channel)
        self.loss_perception = PerceptionLoss(feature_layer=perception_layer, perception_perBand=perception_perBand,
                                              allow_seeded=allow_seeded)

    def forward(self, target_image, generate_image, cmap):
        C = target_image.shape[1]
        generator_loss = _masked_ratio(target_image, generate_image, cmap, 1, True, 1.0 / C, skip_zero=True)
        n = target_image.shape[0]
        z = masked_pair(target_image, generate_image, cmap)
        ssim_loss = 1 - self.ssim(*z.split(n, dim=0))
        perception_loss = self.loss_perception(target_image, generate_image, cmap, stacked=z)
        return generator_loss, ssim_loss, perception_loss


def region_loss(cmap, region, criterion):
    """mean_i[ criterion(cmap_i * region_i, 0) * HW / sum(region_i) ], empty regions
    skipped -- reference Loss.py:127-141.  ``criterion``: nn.L1Loss() or nn.MSELoss()
    instance (as the demos pass) or the strings 'l1' / 'mse'."""
    if isinstance(criterion, str):
        kind = {'l1': 0, 'mse': 1}[criterion]
    elif isinstance(criterion, nn.L1Loss):
        kind = 0
    elif isinstance(criterion, nn.MSELoss):
        kind = 1
    else:
        raise TypeError('region_loss: criterion must be nn.L1Loss / nn.MSELoss')
    return _masked_ratio(cmap, None, region, kind, False, 1.0 / cmap.shape[1], skip_zero=True)
