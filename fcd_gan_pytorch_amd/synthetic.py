"""Synthetic bi-temporal tiles for benchmarks / smoke runs:
standard-normal bands (the reference feeds mean/std-normalised bands,
CommonFunc.py:215), T2 = T1 + 0.1*noise with one seeded rectangle (<= 30 % area)
replaced by fresh noise, region = that rectangle dilated by 10 px as {0,1}
(OSCDProcess.py:41,68-73)."""
import numpy as np
import torch


def synthetic_tiles(seed, N, C, H, W):
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
        region[n, 0, max(0, r0 - 10):min(H, r0 + rh + 10), max(0, c0 - 10):min(W, c0 + rw + 10)] = 1.0
    return torch.from_numpy(x), torch.from_numpy(y), torch.from_numpy(region)
