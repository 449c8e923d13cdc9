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
