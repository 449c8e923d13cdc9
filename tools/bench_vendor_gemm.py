#!/usr/bin/env python3
"""What does the vendor GEMM reach on this chip?  (context for roofline.frac of the split GEMM)

hipBLASLt / rocBLAS through torch.matmul / torch.bmm, HIP events, 20 launches after 5 warm-ups:
  * a large square bf16 GEMM -- the practical matrix-pipe ceiling under the chip's power management (the 2.5 PFLOP/s dense
    bf16 peak assumes 2.4 GHz on every CU);
  * the batched GEMM shapes of the headline step's biggest Winograd launches (36 transform positions, [M x Kc] x [Kc x N]),
    in bf16 (ONE product; the split GEMM executes six per fp32 product) and in fp32 (the library's answer to the same
    fp32-exact problem the split GEMM solves).
Usage: python tools/bench_vendor_gemm.py [--md gpurun_out/r04_vendor_gemm.md]"""
import argparse
import os

import torch


def timeit(fn, n=20, warm=5):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--md', default=None)
    args = ap.parse_args()
    torch.backends.cuda.matmul.allow_tf32 = False
    rows = []
    for n in (4096, 8192, 16384):
        a = torch.randn(n, n, device='cuda', dtype=torch.bfloat16)
        b = torch.randn(n, n, device='cuda', dtype=torch.bfloat16)
        ms = timeit(lambda: torch.matmul(a, b))
        rows.append(('square %d^3' % n, 'bf16', ms, 2.0 * n ** 3 / ms / 1e9, 2500.0))
        del a, b
    # [batch 36] x [M x Kc] x [Kc x N]: conv3_x, conv4_x, conv2_2, conv5_x of the perception VGG at 208 band images
    for tag, M, N, Kc in (('conv3_x', 256, 53248, 256), ('conv4_x', 512, 13312, 512), ('conv2_2', 128, 212992, 128), ('conv5_x', 512, 3328, 512)):
        for dt, name, peak in ((torch.bfloat16, 'bf16', 2500.0), (torch.float32, 'fp32', 157.3)):
            a = torch.randn(36, M, Kc, device='cuda', dtype=dt)
            b = torch.randn(36, Kc, N, device='cuda', dtype=dt)
            c = torch.empty(36, M, N, device='cuda', dtype=dt)
            ms = timeit(lambda: torch.bmm(a, b, out=c))
            rows.append(('%s  36 x [%d x %d] x [%d x %d]' % (tag, M, Kc, Kc, N), name, ms, 2.0 * 36 * M * N * Kc / ms / 1e9, peak))
            del a, b, c
    L = ['# Vendor GEMM on the same chip (tools/bench_vendor_gemm.py; torch %s, hipBLASLt / rocBLAS behind torch.matmul / torch.bmm)' % torch.__version__, '',
         '| GEMM | dtype | ms | TFLOP/s | of the dtype\'s dense MFMA peak |', '|---|---|---|---|---|']
    for tag, name, ms, tf, peak in rows:
        print('%-48s %-5s %8.3f ms  %8.1f TFLOP/s  %.3f of peak' % (tag, name, ms, tf, tf / peak))
        L.append('| %s | %s | %.3f | %.1f | %.3f |' % (tag, name, ms, tf, tf / peak))
    if args.md:
        os.makedirs(os.path.dirname(os.path.abspath(args.md)), exist_ok=True)
        open(args.md, 'w').write('\n'.join(L) + '\n')


if __name__ == '__main__':
    main()
