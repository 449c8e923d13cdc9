"""Do a compute-bound kernel (fused F(2x2) conv, 86 KB LDS, 1 WG/CU) and a bandwidth-bound kernel (ReLU stream / Winograd transforms) from
two HIP streams really run side by side on MI355X?  t(A), t(B), t(A || B)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import fcd_gan_pytorch_amd as p
ops = p._ops
dev = torch.device('cuda', 0)
N = 104
x = torch.randn(N, 64, 256, 256, device=dev)
w = torch.randn(64, 64, 3, 3, device=dev) * 0.05
b = torch.zeros(64, device=dev)
big = torch.randn(N, 64, 256, 256, device=dev)
x2 = torch.randn(N, 128, 128, 128, device=dev)
w2 = torch.randn(128, 128, 3, 3, device=dev) * 0.03
b2 = torch.zeros(128, device=dev)
sa, sb = torch.cuda.Stream(dev), torch.cuda.Stream(dev)

def A():
    return ops.conv2d(x, w, b, 1, 1, relu=True)            # fused F(2x2): compute-bound
def B1():
    return ops.bn_act(big, None, ops.ACT_RELU)              # pure stream, no LDS
def B2():
    return ops.conv2d(x2, w2, b2, 1, 1, relu=True)          # F(4x4) three kernels: transforms + HBM-bound GEMM (whole LDS)

def timed(fa, fb, reps=5):
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(reps):
        t0 = time.perf_counter()
        if fa:
            with torch.cuda.stream(sa):
                for _ in range(3): fa()
        if fb:
            with torch.cuda.stream(sb):
                for _ in range(3): fb()
        torch.cuda.synchronize()
        best = min(best, time.perf_counter() - t0)
    return best * 1e3 / 3

with torch.no_grad():
    for f in (A, B1, B2): f()
    for name, fb in (('relu stream (no LDS)', B1), ('F(4x4) 128ch layer (3 kernels)', B2)):
        ta, tb, tab = timed(A, None), timed(None, fb), timed(A, fb)
        print('%-34s A %.2f ms  B %.2f ms  A||B %.2f ms  (sum %.2f, max %.2f)' % (name, ta, tb, tab, ta + tb, max(ta, tb)))
    # two A's on two streams (same kernel competing)
    ta, taa = timed(A, None), timed(A, A)
    print('A || A: %.2f vs 2 x %.2f' % (taa, ta))
