"""How fast is the stock library path (PyTorch-ROCm -> MIOpen) on the dominant layer shapes?  (reference point only;
MIOpen's kernel search takes ~10 minutes per shape on a fresh box -- run it with a generous timeout, one shape at a time)"""
import sys, time, torch
import torch.nn.functional as F
torch.backends.cudnn.benchmark = True
shapes = [(208, 256, 64, 256), (208, 64, 256, 64), (208, 512, 32, 512), (8, 1024, 64, 512), (16, 128, 128, 128)]
for (N, C, H, K) in shapes:
    x = torch.randn(N, C, H, H, device='cuda', requires_grad=True)
    w = torch.randn(K, C, 3, 3, device='cuda', requires_grad=True) * 0.05
    b = torch.zeros(K, device='cuda')
    gy = torch.randn(N, K, H, H, device='cuda')
    flops = 2.0 * N * K * H * H * C * 9
    def fwd():
        return F.conv2d(x, w, b, 1, 1)
    for _ in range(3):
        y = fwd()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        y = fwd()
    e1.record(); torch.cuda.synchronize()
    tf = e0.elapsed_time(e1) / 5
    y = fwd()
    for _ in range(2):
        gx, gw = torch.autograd.grad(y, (x, w), gy, retain_graph=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(3):
        gx, gw = torch.autograd.grad(y, (x, w), gy, retain_graph=True)
    e1.record(); torch.cuda.synchronize()
    tb = e0.elapsed_time(e1) / 3
    print('N%d C%d H%d K%d  fwd %.2f ms (%.0f TF)  dgrad+wgrad %.2f ms (%.0f TF)' % (N, C, H, K, tf, flops / tf / 1e9, tb, 2 * flops / tb / 1e9), flush=True)
