#!/usr/bin/env python3
"""Accuracy of the three-kernel Winograd path with the batched GEMM on (a) v_mfma_f32_32x32x2_f32 and (b) the bf16
matrix pipe with exact three-way operand splitting, both against an fp64 convolution (CPU)."""
import os
import sys

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from fcd_gan_pytorch_amd import _ops as ops          # noqa: E402
from fcd_gan_pytorch_amd._lib import lib             # noqa: E402

CASES = [(2, 256, 64, 256), (2, 128, 96, 128), (1, 512, 32, 512), (3, 160, 40, 192), (2, 256, 64, 384), (1, 512, 36, 512), (1, 96, 52, 320)]


def main():
    for N, C, HW, K in CASES:
        g = torch.Generator().manual_seed(C + K)
        x = torch.randn(N, C, HW, HW, generator=g)
        w = torch.randn(K, C, 3, 3, generator=g) * (2.0 / (9 * C)) ** 0.5
        b = torch.randn(K, generator=g) * 0.1
        ref = F.conv2d(x.double(), w.double(), b.double(), padding=1)
        scale = ref.abs().max().item()
        out = {}
        for split in (0, 1):
            lib.fcd_conv_wino_split_set(split)
            xd, wd, bd = x.cuda(), w.cuda().clone(), b.cuda()
            with torch.no_grad():
                y = ops.conv2d(xd, wd, bd, 1, 1)
            out[split] = y.double().cpu()
        e0 = (out[0] - ref).abs()
        e1 = (out[1] - ref).abs()
        print('N=%d C=%d HW=%d K=%d  |ref|max %.3f   f32-mfma: max %.3e rms %.3e    bf16x6: max %.3e rms %.3e   split-vs-f32 max %.3e'
              % (N, C, HW, K, scale, e0.max(), e0.pow(2).mean().sqrt(), e1.max(), e1.pow(2).mean().sqrt(),
                 (out[0] - out[1]).abs().max()))


if __name__ == '__main__':
    main()
