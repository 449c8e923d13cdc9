"""Deterministic weights and synthetic bi-temporal tiles for the parity tests.

Shared by ``gen_golden.py`` (which runs the imported reference on them, in the
build container only) and by the tests (which re-generate the same inputs and
compare the oracle / the HIP path against the stored outputs).  NumPy PCG64
only -- independent of torch's RNG stream.
"""
import numpy as np
import torch


def seeded_state(spec, seed):
    """name->shape spec  =>  name->tensor, reproducible per (seed, position)."""
    sd = {}
    for i, (name, shape) in enumerate(spec.items()):
        rng = np.random.default_rng([seed, i])
        if name.endswith('num_batches_tracked'):
            sd[name] = torch.zeros((), dtype=torch.int64)
            continue
        n = rng.standard_normal(shape).astype(np.float32)
        if len(shape) == 4:                                   # conv / convT weight
            fan_in = shape[1] * shape[2] * shape[3]
            v = n * np.float32(np.sqrt(2.0 / fan_in))
        elif name.endswith('running_mean'):
            v = 0.1 * n
        elif name.endswith('running_var'):
            v = 1.0 + 0.25 * np.tanh(n)
        elif name.endswith('.weight') and tuple(shape) == (1,):   # PReLU slope
            v = 0.25 + 0.05 * np.tanh(n)
        elif name.endswith('.weight'):                        # BN gamma
            v = 1.0 + 0.1 * n
        else:                                                 # biases / BN beta
            v = 0.05 * n
        sd[name] = torch.from_numpy(np.ascontiguousarray(v, dtype=np.float32))
    return sd


def seeded_tiles(seed, N, C, H, W):
    """x ~ N(0,1) bands (the reference feeds mean/std-normalised
    bands, CommonFunc.py:215); y = x + 0.1 noise with one seeded rectangle
    (<=30% area) replaced by fresh noise; region = rectangle dilated by 10 px
    (OSCDProcess.py:41,68-73) as {0,1} float."""
    rng = np.random.default_rng(seed)
    x = rng.standard_normal((N, C, H, W)).astype(np.float32)
    y = (x + 0.1 * rng.standard_normal((N, C, H, W))).astype(np.float32)
    region = np.zeros((N, 1, H, W), np.float32)
    for n in range(N):
        rh = int(rng.integers(max(2, H // 8), max(3, H // 2)))
        rw = int(rng.integers(max(2, W // 8), max(3, int(0.6 * W))))
        r0 = int(rng.integers(0, H - rh + 1))
        c0 = int(rng.integers(0, W - rw + 1))
        y[n, :, r0:r0 + rh, c0:c0 + rw] = rng.standard_normal((C, rh, rw)).astype(np.float32)
        d = min(10, max(1, H // 16))
        region[n, 0, max(0, r0 - d):min(H, r0 + rh + d), max(0, c0 - d):min(W, c0 + rw + d)] = 1.0
    return torch.from_numpy(x), torch.from_numpy(y), torch.from_numpy(region)


def summary(t, nsamp=8, seed=7):
    """Compact fingerprint of a tensor: (sum, L2, nsamp sampled elements)."""
    a = t.detach().double().reshape(-1).numpy()
    idx = np.random.default_rng([seed, a.size]).integers(0, a.size, nsamp) if a.size else np.zeros(0, int)
    return np.concatenate([[a.sum(), np.sqrt((a * a).sum())], a[idx]]).astype(np.float64)


# ------------------------------------------------------------------ ImageNet-LIKE VGG16 statistics (no checkpoint offline)
# Post-ReLU activation rms the calibrated stack reaches per conv layer, input bands ~ N(0, 1): the growth trained VGG16
# features show (activations of order 1 after conv1_x rising to 10^2 rms / 10^3 peaks by conv5_x), which is what stresses the
# F(4x4) transforms -- their rounding scales with the dynamic range inside a 6 x 6 patch.  The He-initialised seeded stack of
# the other tests keeps every layer at rms ~1.
VGG_LIKE_RMS = (1.5, 3.0, 5.0, 8.0, 12.0, 18.0, 25.0, 35.0, 50.0, 70.0, 90.0, 110.0, 130.0)
# mean of each layer's biases in units of its pre-activation std (trained VGG biases are not centred: conv1_1's are ~ +0.5 with
# inputs of unit variance, the deep layers' drift positive as well), and their spread in the same units
VGG_LIKE_BIAS = ((0.5, 0.35), (0.05, 0.3), (0.05, 0.2), (0.05, 0.2), (0.03, 0.15), (0.03, 0.15), (0.05, 0.15), (0.02, 0.1),
                 (0.03, 0.1), (0.05, 0.12), (0.05, 0.12), (0.08, 0.15), (0.12, 0.2))


def imagenet_like_vgg_state(spec, seed, probe):
    """A seeded VGG16 ``features`` state with the qualitative statistics of the ImageNet checkpoint the reference loads
    (Loss.py:25) -- which cannot be had offline: HEAVY-TAILED filters (Student-t, 4 degrees of freedom, instead of Gaussian
    draws; a slightly negative mean, so activations are sparse), biases with a non-zero mean, and layer gains calibrated on the
    (3, H, W) ``probe`` image in fp64 so that the post-ReLU activation rms follows ``VGG_LIKE_RMS`` (1.5 ... 130).  Returns
    (state dict fp32, per-layer post-ReLU rms / max measured on the probe in fp64).  Deterministic per (seed, probe)."""
    import torch.nn.functional as F
    names = [k[:-len('.weight')] for k in spec if k.endswith('.weight')]
    sd, report = {}, []
    z = probe.double().unsqueeze(0)
    for li, name in enumerate(names):
        shape = spec[name + '.weight']
        rng = np.random.default_rng([seed, li])
        w = rng.standard_t(4, size=shape)
        w = (w - 0.05 * w.std()) / w.std()                         # unit std, mean -0.05 std
        w = torch.from_numpy(w)
        pre = F.conv2d(z, w, None, padding=1)
        s = pre.std().item()
        bm, bs = VGG_LIKE_BIAS[li]
        b = torch.from_numpy(rng.standard_normal(shape[0])) * (bs * s) + bm * s
        act = torch.relu(pre + b.view(1, -1, 1, 1))
        gain = VGG_LIKE_RMS[li] / act.pow(2).mean().sqrt().item()
        w32, b32 = (w * gain).float(), (b * gain).float()
        sd[name + '.weight'], sd[name + '.bias'] = w32, b32
        z = torch.relu(F.conv2d(z, w32.double(), b32.double(), padding=1))
        report.append((name, z.pow(2).mean().sqrt().item(), z.max().item(), (z > 0).double().mean().item()))
        if li in (1, 3, 6, 9, 12):                                  # pools of torchvision's cfg D
            z = F.max_pool2d(z, 2)
    return sd, report
