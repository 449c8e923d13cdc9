"""Oracle (test infrastructure): functional forward passes of the three FCD-GAN
networks over a flat ``state_dict`` (same key names as the reference's
``nn.Module``s, so reference checkpoints and seeded fixtures plug in directly).

Every function cites the reference lines it restates.  Pure torch CPU ops.
"""
import torch
import torch.nn.functional as F

BN_EPS = 1e-5       # nn.BatchNorm2d default used everywhere in Module.py
BN_MOMENTUM = 0.1


def _bn(sd, key, x, train):
    """nn.BatchNorm2d (Module.py:26,30,156,178,181,199,203,207): batch
    statistics + running-stat update in train mode, running stats in eval."""
    if train:
        sd[key + '.num_batches_tracked'] += 1
    return F.batch_norm(x, sd[key + '.running_mean'], sd[key + '.running_var'],
                        sd[key + '.weight'], sd[key + '.bias'],
                        training=train, momentum=BN_MOMENTUM, eps=BN_EPS)


def _conv(sd, key, x, stride=1, padding=0):
    return F.conv2d(x, sd[key + '.weight'], sd[key + '.bias'], stride=stride, padding=padding)


def double_conv(sd, pre, x, train):
    """DoubleConv, Module.py:18-35: (conv3x3 p1 -> BN -> ReLU) x 2; the
    Sequential indices 0,1 and 3,4 carry the parameters."""
    x = F.relu(_bn(sd, pre + '.double_conv.1', _conv(sd, pre + '.double_conv.0', x, padding=1), train))
    x = F.relu(_bn(sd, pre + '.double_conv.4', _conv(sd, pre + '.double_conv.3', x, padding=1), train))
    return x


def down(sd, pre, x, train):
    """Down, Module.py:38-49: MaxPool2d(2) then DoubleConv (Sequential idx 1)."""
    return double_conv(sd, pre + '.maxpool_conv.1', F.max_pool2d(x, 2), train)


def up(sd, pre, x1, x2, train, bilinear):
    """Up, Module.py:52-79: x2 upsample (bilinear align_corners=True, or
    ConvTranspose2d k2 s2), zero-pad to the skip's size, cat([skip, up]),
    DoubleConv."""
    if bilinear:
        x1 = F.interpolate(x1, scale_factor=2, mode='bilinear', align_corners=True)
    else:
        x1 = F.conv_transpose2d(x1, sd[pre + '.up.weight'], sd[pre + '.up.bias'], stride=2)
    dy = x2.shape[2] - x1.shape[2]
    dx = x2.shape[3] - x1.shape[3]
    x1 = F.pad(x1, [dx // 2, dx - dx // 2, dy // 2, dy - dy // 2])
    return double_conv(sd, pre + '.conv', torch.cat([x2, x1], dim=1), train)


def segmentor(sd, x1, x2, train=True, bilinear=True):
    """Segmentor.forward, Module.py:113-140.  Siamese encoder: every encoder
    stage is applied to x1 THEN x2 (two BN-stat updates, in that order)."""
    feats = []
    a, b = double_conv(sd, 'inc', x1, train), double_conv(sd, 'inc', x2, train)
    feats.append(torch.cat([a, b], dim=1))
    for stage in ('down1', 'down2', 'down3', 'down4'):
        a = down(sd, stage, a, train)
        b = down(sd, stage, b, train)
        feats.append(torch.cat([a, b], dim=1))
    x = up(sd, 'up1', feats[4], feats[3], train, bilinear)
    x = up(sd, 'up2', x, feats[2], train, bilinear)
    x = up(sd, 'up3', x, feats[1], train, bilinear)
    x = up(sd, 'up4', x, feats[0], train, bilinear)
    # OutConv, Module.py:82-90: 1x1 conv + sigmoid -> change-density map
    return torch.sigmoid(_conv(sd, 'outc.conv', x))


def residual_block(sd, pre, x, train):
    """ResidualBlock, Module.py:174-190."""
    r = _conv(sd, pre + '.conv1', x, padding=1)
    r = F.prelu(_bn(sd, pre + '.bn1', r, train), sd[pre + '.prelu.weight'])
    r = _bn(sd, pre + '.bn2', _conv(sd, pre + '.conv2', r, padding=1), train)
    return x + r


def generator(sd, x, train=True):
    """Generator.forward, Module.py:160-172 (raw output, no tanh)."""
    b1 = F.prelu(_conv(sd, 'block1.0', x, padding=4), sd['block1.1.weight'])
    h = b1
    for i in range(2, 7):
        h = residual_block(sd, 'block%d' % i, h, train)
    h = _bn(sd, 'block7.1', _conv(sd, 'block7.0', h, padding=1), train)
    return _conv(sd, 'block8', b1 + h, padding=4)


def _disc_net(sd, x, train):
    """Discriminator_SRGAN_simple.net, Module.py:195-209."""
    x = F.leaky_relu(_conv(sd, 'net.0', x, stride=2, padding=1), 0.2)
    for ci, bi in ((2, 3), (5, 6), (8, 9)):
        x = _conv(sd, 'net.%d' % ci, x, stride=2, padding=1)
        x = F.leaky_relu(_bn(sd, 'net.%d' % bi, x, train), 0.2)
    return x


def discriminator(sd, x, y, train=True):
    """Discriminator_SRGAN_simple.forward, Module.py:219-223: shared net on x
    then y, classifier on the difference, sigmoid, flatten to (N,)."""
    fx = _disc_net(sd, x, train)
    fy = _disc_net(sd, y, train)
    d = F.adaptive_avg_pool2d(fx - fy, 1)
    d = F.leaky_relu(_conv(sd, 'classifier.1', d), 0.2)
    d = _conv(sd, 'classifier.3', d)
    return torch.sigmoid(d.view(x.shape[0]))


# ----------------------------------------------------------------------------
# VGG16 ``features`` stack (torchvision cfg D), as used by Loss.py:25-36.
VGG_CFG = [64, 64, 'M', 128, 128, 'M', 256, 256, 256, 'M', 512, 512, 512, 'M', 512, 512, 512, 'M']


def vgg_layers():
    """[(index, kind, cin, cout)] for the 31 entries of vgg16().features."""
    out, idx, cin = [], 0, 3
    for v in VGG_CFG:
        if v == 'M':
            out.append((idx, 'pool', cin, cin)); idx += 1
        else:
            out.append((idx, 'conv', cin, v)); idx += 1
            out.append((idx, 'relu', v, v)); idx += 1
            cin = v
    return out


def vgg_features(sd, x, taps, prefix=''):
    """Run all 31 layers (Loss.py:45-49 runs the whole stack even past the last
    tapped index) and return the activations at ``taps`` in layer order."""
    got = {}
    for idx, kind, _, _ in vgg_layers():
        if kind == 'conv':
            x = F.conv2d(x, sd['%s%d.weight' % (prefix, idx)], sd['%s%d.bias' % (prefix, idx)], padding=1)
        elif kind == 'relu':
            x = F.relu(x)
        else:
            x = F.max_pool2d(x, 2)
        if idx in taps:
            got[idx] = x
    return got


# ----------------------------------------------------------------------------
def clone_state(sd, requires_grad=True):
    """Detached float copies; parameters (not BN buffers) get requires_grad."""
    out = {}
    for k, v in sd.items():
        t = v.detach().clone()
        if requires_grad and t.is_floating_point() and 'running_' not in k:
            t.requires_grad_(True)
        out[k] = t
    return out


def param_keys(sd):
    return [k for k, v in sd.items() if v.is_floating_point() and 'running_' not in k]


# ----------------------------------------------------------------------------
# state_dict layouts (name -> shape), in the reference modules' registration
# order.  Pinned against the imported reference by tests/golden/gen_golden.py.
def _conv_spec(out, name, cout, cin, k):
    out[name + '.weight'] = (cout, cin, k, k)
    out[name + '.bias'] = (cout,)


def _bn_spec(out, name, c):
    out[name + '.weight'] = (c,)
    out[name + '.bias'] = (c,)
    out[name + '.running_mean'] = (c,)
    out[name + '.running_var'] = (c,)
    out[name + '.num_batches_tracked'] = ()


def _dc_spec(out, pre, cin, cout, mid=None):
    mid = mid or cout
    _conv_spec(out, pre + '.double_conv.0', mid, cin, 3)
    _bn_spec(out, pre + '.double_conv.1', mid)
    _conv_spec(out, pre + '.double_conv.3', cout, mid, 3)
    _bn_spec(out, pre + '.double_conv.4', cout)


def segmentor_spec(n_channels, n_out=1, bilinear=True):
    """Key/shape list of Segmentor (Module.py:94-111)."""
    o = {}
    f = 2 if bilinear else 1
    _dc_spec(o, 'inc', n_channels, 64)
    for name, cin, cout in (('down1', 64, 128), ('down2', 128, 256), ('down3', 256, 512),
                            ('down4', 512, 1024 // f)):
        _dc_spec(o, name + '.maxpool_conv.1', cin, cout)
    for name, cin, cout in (('up1', 2048, 1024 // f), ('up2', 1024, 512 // f),
                            ('up3', 512, 256 // f), ('up4', 256, 128)):
        if bilinear:
            _dc_spec(o, name + '.conv', cin, cout, cin // 2)
        else:
            o[name + '.up.weight'] = (cin, cin // 2, 2, 2)
            o[name + '.up.bias'] = (cin // 2,)
            _dc_spec(o, name + '.conv', cin, cout)
    _conv_spec(o, 'outc.conv', n_out, 128, 1)
    return o


def generator_spec(n_channels):
    """Key/shape list of Generator (Module.py:143-158)."""
    o = {}
    _conv_spec(o, 'block1.0', 64, n_channels, 9)
    o['block1.1.weight'] = (1,)
    for i in range(2, 7):
        p = 'block%d' % i
        _conv_spec(o, p + '.conv1', 64, 64, 3)
        _bn_spec(o, p + '.bn1', 64)
        o[p + '.prelu.weight'] = (1,)
        _conv_spec(o, p + '.conv2', 64, 64, 3)
        _bn_spec(o, p + '.bn2', 64)
    _conv_spec(o, 'block7.0', 64, 64, 3)
    _bn_spec(o, 'block7.1', 64)
    _conv_spec(o, 'block8', n_channels, 64, 9)
    return o


def discriminator_spec(n_channels=3):
    """Key/shape list of Discriminator_SRGAN_simple (Module.py:193-217)."""
    o = {}
    _conv_spec(o, 'net.0', 64, n_channels, 3)
    for ci, bi, cin, cout in ((2, 3, 64, 128), (5, 6, 128, 256), (8, 9, 256, 512)):
        _conv_spec(o, 'net.%d' % ci, cout, cin, 3)
        _bn_spec(o, 'net.%d' % bi, cout)
    _conv_spec(o, 'classifier.1', 1024, 512, 1)
    _conv_spec(o, 'classifier.3', 1, 1024, 1)
    return o


def vgg_spec(prefix=''):
    o = {}
    for idx, kind, cin, cout in vgg_layers():
        if kind == 'conv':
            _conv_spec(o, '%s%d' % (prefix, idx), cout, cin, 3)
    return o
