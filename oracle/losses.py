"""Oracle (test infrastructure): loss terms of the FCD-GAN train step, restated
functionally on torch CPU ops.  Reference: Loss.py, ssim.py, and the inline
adversarial terms of Demo_RSSS.py / Demo_WSSS.py.
"""
import torch
import torch.nn.functional as F

from . import nets

MS_WEIGHTS = (0.0448, 0.2856, 0.3001, 0.2363, 0.1333)   # ssim.py:200


# ------------------------------------------------------------------ MS-SSIM
def gauss_window(size=11, sigma=1.5):
    """ssim.py:9-23 -- normalised 1-D Gaussian taps, float32."""
    c = torch.arange(size).to(dtype=torch.float)
    c -= size // 2
    g = torch.exp(-(c ** 2) / (2 * sigma ** 2))
    g /= g.sum()
    return g


def blur_valid(x, g):
    """ssim.py:26-52 -- separable depth-wise VALID correlation, H then W; a
    dimension shorter than the window is skipped."""
    C = x.shape[1]
    k = g.numel()
    if x.shape[2] >= k:
        x = F.conv2d(x, g.view(1, 1, k, 1).repeat(C, 1, 1, 1), stride=1, padding=0, groups=C)
    if x.shape[3] >= k:
        x = F.conv2d(x, g.view(1, 1, 1, k).repeat(C, 1, 1, 1), stride=1, padding=0, groups=C)
    return x


def ssim_level(X, Y, g, data_range=1.0, K=(0.01, 0.03)):
    """ssim.py:55-92 -- per-(n,c) spatial means of ssim_map and cs_map."""
    C1 = (K[0] * data_range) ** 2
    C2 = (K[1] * data_range) ** 2
    mu1, mu2 = blur_valid(X, g), blur_valid(Y, g)
    mu1_sq, mu2_sq, mu12 = mu1.pow(2), mu2.pow(2), mu1 * mu2
    s1 = 1.0 * (blur_valid(X * X, g) - mu1_sq)
    s2 = 1.0 * (blur_valid(Y * Y, g) - mu2_sq)
    s12 = 1.0 * (blur_valid(X * Y, g) - mu12)
    cs_map = (2 * s12 + C2) / (s1 + s2 + C2)
    ssim_map = ((2 * mu12 + C1) / (mu1_sq + mu2_sq + C1)) * cs_map
    return torch.flatten(ssim_map, 2).mean(-1), torch.flatten(cs_map, 2).mean(-1)


def ms_ssim(X, Y, data_range=1.0, size_average=True, win_size=11, win_sigma=1.5,
            weights=None, K=(0.01, 0.03)):
    """ssim.py:153-225 -- 5-level MS-SSIM; avg_pool2d(k2, padding=s%2) between
    levels; relu on cs (levels 0..3) and on ssim (level 4); prod of powers."""
    if X.shape != Y.shape:
        raise ValueError("Input images should have the same dimensions.")
    if min(X.shape[-2:]) <= (win_size - 1) * 16:
        raise AssertionError("Image size should be larger than %d" % ((win_size - 1) * 16))
    w = torch.tensor(list(weights or MS_WEIGHTS), dtype=X.dtype, device=X.device)
    g = gauss_window(win_size, win_sigma).to(dtype=X.dtype, device=X.device)      # (device: the fp64 truth of the full-size tests may run on the GPU's stock fp64 ops)
    vals = []
    L = w.numel()
    for lvl in range(L):
        s, cs = ssim_level(X, Y, g, data_range, K)
        if lvl < L - 1:
            vals.append(torch.relu(cs))
            pad = [d % 2 for d in X.shape[2:]]
            X = F.avg_pool2d(X, kernel_size=2, padding=pad)
            Y = F.avg_pool2d(Y, kernel_size=2, padding=pad)
    vals.append(torch.relu(s))
    stack = torch.stack(vals, dim=0)
    v = torch.prod(stack ** w.view(-1, 1, 1), dim=0)
    return v.mean() if size_average else v.mean(1)


# --------------------------------------------------------------- perception
TAP_ORDER = (29, 22, 15, 8, 3)     # Loss.py:30
# Test-infrastructure switch: run the per-band VGG passes of Loss.py:50-60 as ONE batch of C x N band images instead of C
# sequential passes of N.  Every band is an independent sample, and sum_b mse(fx_b, fy_b) / C over equally sized tensors is the mse
# over their concatenation, so the value and gradients are the same function; only the fp64 TRUTH runs of the full-size tests
# switch it on (torch's double-precision CPU convolution parallelises over the batch only: 2 of 16 cores busy otherwise).
BATCH_BANDS = False


def perception(vgg_sd, target, generated, cmask, feature_layer=1, per_band=False, prefix=''):
    """PerceptionLoss.forward, Loss.py:38-61."""
    nl = min(max(feature_layer, 1), 5)
    taps = TAP_ORDER[:nl]
    total = 0
    if not per_band:
        assert target.shape[1] >= 3
        m = 1 - cmask.repeat((1, 3, 1, 1))
        fx = nets.vgg_features(vgg_sd, target[:, 0:3] * m, taps, prefix)
        fy = nets.vgg_features(vgg_sd, generated[:, 0:3] * m, taps, prefix)
        for i in sorted(taps):
            total = total + F.mse_loss(fx[i], fy[i]) / nl
    elif BATCH_BANDS:
        n, C, H, W = target.shape
        keep = 1 - cmask
        xb = (target * keep).reshape(n * C, 1, H, W).repeat((1, 3, 1, 1))
        yb = (generated * keep).reshape(n * C, 1, H, W).repeat((1, 3, 1, 1))
        fx = nets.vgg_features(vgg_sd, xb, taps, prefix)
        fy = nets.vgg_features(vgg_sd, yb, taps, prefix)
        for i in sorted(taps):
            total = total + F.mse_loss(fx[i], fy[i]) / nl
    else:
        C = target.shape[1]
        for b in range(C):
            xb = (target[:, b].unsqueeze(1) * (1 - cmask)).repeat((1, 3, 1, 1))
            yb = (generated[:, b].unsqueeze(1) * (1 - cmask)).repeat((1, 3, 1, 1))
            fx = nets.vgg_features(vgg_sd, xb, taps, prefix)
            fy = nets.vgg_features(vgg_sd, yb, taps, prefix)
            for i in sorted(taps):
                total = total + F.mse_loss(fx[i], fy[i]) / nl / C
    return total


# ---------------------------------------------------------- reconstruction
def _masked_pair(target, generated, cmap):
    C = target.shape[1]
    keep = 1 - cmap.repeat((1, C, 1, 1))
    return target * keep, generated * keep


def cnet_loss(vgg_sd, target, generated, cmap, mask_switch=False, perception_layer=1,
              per_band=True, vgg_prefix=''):
    """CNetLoss.forward, Loss.py:73-95 (USSS).  Returns
    (generator_loss, l1_loss, perception_loss, ssim_loss).  L1 reconstruction,
    NO guard against an all-changed sample."""
    cmask = (torch.sign(cmap - 0.5) + 1) / 2
    npx = target.shape[2] * target.shape[3]
    wnc = torch.sum(1 - cmap, (1, 2, 3))
    tm, gm = _masked_pair(target, generated, cmap)
    rec = 0
    for i in range(target.shape[0]):
        rec = rec + F.l1_loss(tm[i], gm[i]) * npx / wnc[i]
    rec = rec / target.shape[0]
    l1 = torch.mean(abs(cmap))
    perc = perception(vgg_sd, target, generated, cmask if mask_switch else cmap,
                      perception_layer, per_band, vgg_prefix)
    ssim_loss = 1 - ms_ssim(tm, gm, data_range=1.0)
    return rec, l1, perc, ssim_loss


def cgenerator_loss(vgg_sd, target, generated, cmap, perception_layer=1, per_band=False,
                    vgg_prefix=''):
    """CGeneratorLoss.forward, Loss.py:108-124 (WSSS/RSSS).  Returns
    (generator_loss, ssim_loss, perception_loss).  MSE reconstruction; samples
    whose keep-weight sum is exactly 0 are skipped."""
    npx = target.shape[2] * target.shape[3]
    wnc = torch.sum(1 - cmap, (1, 2, 3))
    tm, gm = _masked_pair(target, generated, cmap)
    rec = 0
    for i in range(target.shape[0]):
        if wnc[i] == 0:
            continue
        rec = rec + F.mse_loss(tm[i], gm[i]) * npx / wnc[i]
    rec = rec / target.shape[0]
    ssim_loss = 1 - ms_ssim(tm, gm, data_range=1.0)
    perc = perception(vgg_sd, target, generated, cmap, perception_layer, per_band, vgg_prefix)
    return rec, ssim_loss, perc


def region_loss(cmap, region, kind):
    """region_loss, Loss.py:127-141; ``kind`` in {'l1','mse'} stands for the
    nn.L1Loss()/nn.MSELoss() criterion argument."""
    fn = F.l1_loss if kind == 'l1' else F.mse_loss
    npx = cmap.shape[2] * cmap.shape[3]
    nreg = torch.sum(region, (1, 2, 3))
    masked = cmap * region
    zero = torch.zeros_like(region)
    acc = 0
    for i in range(cmap.shape[0]):
        if nreg[i] == 0:
            continue
        acc = acc + fn(masked[i], zero[i]) * npx / nreg[i]
    return acc / cmap.shape[0]
