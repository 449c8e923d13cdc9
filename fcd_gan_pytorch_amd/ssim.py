"""Drop-in for the reference's ``ssim.py`` public surface -- ``ssim``, ``ms_ssim``,
``SSIM``, ``MS_SSIM`` with the same arguments, defaults and exceptions -- on the
fused HIP SSIM-level kernel (csrc/ssim.hip): per level ONE kernel reads the X,Y
tile, forms the five Gaussian-window statistics in LDS and reduces the ssim/cs
maps per (n,c); the 2x2 padded average pool between levels is a second small
kernel.  Only the O(levels*N*C) tail (relu, powers, product, mean) uses torch ops.
2-D images (N,C,H,W), odd window <= 11 taps, CUDA/ROCm tensors only.

Deliberate differences from the vendored pytorch-msssim (none is on a demo's path; each RAISES instead of
silently computing something else):
 * 5-D (N,C,T,H,W) inputs (``conv3d`` branch, ssim.py:120-137,171-186) -> ValueError: 4-d tensors only;
 * windows longer than 11 taps -> error.
``gaussian_filter``'s "skip a spatial dimension shorter than the window, with a warning" (ssim.py:44-50) is reproduced [r4]: the
kernel applies the single tap 1 along such a dimension (``ssim`` only -- MS-SSIM's own assert, ssim.py:194-197, requires
min(H, W) > 160 for the default window).
"""
import warnings

import torch

from . import _ops as ops

_DEFAULT_WEIGHTS = (0.0448, 0.2856, 0.3001, 0.2363, 0.1333)   # reference ssim.py:200


def _fspecial_gauss_1d(size, sigma):
    """1-D Gaussian taps as a (1,1,size) float tensor (reference ssim.py:9-23)."""
    offs = torch.arange(size, dtype=torch.float32) - (size // 2)
    g = torch.exp(-(offs * offs) / (2 * sigma ** 2))      # host-side, 11 numbers, fp32 like the reference
    return (g / g.sum()).view(1, 1, size)


_DEV_CONST = {}


def _dev_const(values, device):
    """A small constant vector on the device, uploaded ONCE per (values, device): the window taps and the level weights are the
    same 11 + 5 numbers in every call, and a host-to-device copy per call is both two launches per step and the one thing in the
    loss that a hipGraph capture of the train step cannot record (a caller's own ``torch.cuda.graph``)."""
    key = (tuple(float(v) for v in values), str(device))
    t = _DEV_CONST.get(key)
    if t is None:
        t = _DEV_CONST[key] = torch.tensor(key[0], dtype=torch.float32, device=device)
    return t


def _taps_from(win, device):
    """Accept the reference's (C,1,1,k) / (1,1,k) window tensors; return the k taps."""
    w = win.reshape(-1, win.shape[-1])[0]
    if w.is_cuda:
        return w.to(device=device, dtype=torch.float32).contiguous()
    return _dev_const(w.tolist(), device)


def _check_pair(X, Y):
    if not X.shape == Y.shape:
        raise ValueError("Input images should have the same dimensions.")
    for d in range(len(X.shape) - 1, 1, -1):
        X = X.squeeze(dim=d)
        Y = Y.squeeze(dim=d)
    if not X.type() == Y.type():
        raise ValueError("Input images should have the same dtype.")
    return X, Y


def _consts(data_range, K):
    return (K[0] * data_range) ** 2, (K[1] * data_range) ** 2


def ssim(X, Y, data_range=255, size_average=True, win_size=11, win_sigma=1.5, win=None, K=(0.01, 0.03),
         nonnegative_ssim=False):
    """Single-scale SSIM (reference ssim.py:95-150)."""
    X, Y = _check_pair(X, Y)
    if len(X.shape) != 4:
        raise ValueError(f"Input images should be 4-d tensors, but got {X.shape}")
    if win is not None:
        win_size = win.shape[-1]
    if not (win_size % 2 == 1):
        raise ValueError("Window size should be odd.")
    taps = _taps_from(win if win is not None else _fspecial_gauss_1d(win_size, win_sigma), X.device)
    for i, sdim in enumerate(X.shape[2:]):            # the reference's warning, from gaussian_filter (ssim.py:46-50)
        if sdim < win_size:
            warnings.warn(f"Skipping Gaussian Smoothing at dimension 2+{i} for input: {X.shape} and win size: {win_size}")
    C1, C2 = _consts(data_range, K)
    per_channel, _ = ops.ssim_level(X, Y, taps, C1, C2)
    if nonnegative_ssim:
        per_channel = torch.relu(per_channel)
    return per_channel.mean() if size_average else per_channel.mean(1)


def ms_ssim(X, Y, data_range=255, size_average=True, win_size=11, win_sigma=1.5, win=None, weights=None,
            K=(0.01, 0.03)):
    """Multi-scale SSIM (reference ssim.py:153-225): 5 levels, avg_pool2d(2, padding=s%2)
    between levels, relu on cs (levels 0-3) / ssim (level 4), product of powers."""
    X, Y = _check_pair(X, Y)
    if len(X.shape) != 4:
        raise ValueError(f"Input images should be 4-d tensors, but got {X.shape}")
    if win is not None:
        win_size = win.shape[-1]
    if not (win_size % 2 == 1):
        raise ValueError("Window size should be odd.")
    smaller_side = min(X.shape[-2:])
    assert smaller_side > (win_size - 1) * (2 ** 4), \
        "Image size should be larger than %d due to the 4 downsamplings in ms-ssim" % ((win_size - 1) * (2 ** 4))
    w = _dev_const(list(weights) if weights is not None else list(_DEFAULT_WEIGHTS), X.device).to(X.dtype)
    taps = _taps_from(win if win is not None else _fspecial_gauss_1d(win_size, win_sigma), X.device)
    C1, C2 = _consts(data_range, K)
    levels = w.shape[0]
    terms = []
    for lvl in range(levels):
        s_c, cs = ops.ssim_level(X, Y, taps, C1, C2)
        if lvl < levels - 1:
            terms.append(torch.relu(cs))
            X = ops.avgpool2_pad(X)
            Y = ops.avgpool2_pad(Y)
    terms.append(torch.relu(s_c))
    # prod_l term_l ** w_l (reference ssim.py:219-222: torch.prod over the stacked levels).  Written as a chain of products:
    # torch.prod's backward counts the zeros of its input with ``.item()`` -- a host synchronisation in the middle of every
    # backward pass that runs through MS-SSIM (the USSS steps), and not capturable in a hipGraph.  The chain's gradient w.r.t. a
    # level is the product of the other levels, which is what prod_backward computes on its zero-safe path.
    val = terms[0] ** w[0]
    for lvl in range(1, levels):
        val = val * terms[lvl] ** w[lvl]
    return val.mean() if size_average else val.mean(1)


class SSIM(torch.nn.Module):
    def __init__(self, data_range=255, size_average=True, win_size=11, win_sigma=1.5, channel=3, spatial_dims=2,
                 K=(0.01, 0.03), nonnegative_ssim=False):
        super(SSIM, self).__init__()
        self.win_size = win_size
        self.win = _fspecial_gauss_1d(win_size, win_sigma).repeat([channel, 1] + [1] * spatial_dims)
        self.size_average = size_average
        self.data_range = data_range
        self.K = K
        self.nonnegative_ssim = nonnegative_ssim

    def forward(self, X, Y):
        return ssim(X, Y, data_range=self.data_range, size_average=self.size_average, win=self.win, K=self.K,
                    nonnegative_ssim=self.nonnegative_ssim)


class MS_SSIM(torch.nn.Module):
    def __init__(self, data_range=255, size_average=True, win_size=11, win_sigma=1.5, channel=3, spatial_dims=2,
                 weights=None, K=(0.01, 0.03)):
        super(MS_SSIM, self).__init__()
        self.win_size = win_size
        self.win = _fspecial_gauss_1d(win_size, win_sigma).repeat([channel, 1] + [1] * spatial_dims)
        self.size_average = size_average
        self.data_range = data_range
        self.weights = weights
        self.K = K

    def forward(self, X, Y):
        return ms_ssim(X, Y, data_range=self.data_range, size_average=self.size_average, win=self.win,
                       weights=self.weights, K=self.K)
