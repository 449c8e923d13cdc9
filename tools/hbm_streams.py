import torch
def t(fn, it=20):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(it): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / it
n = 1 << 30
x = torch.empty(n, device='cuda'); y = torch.empty(n, device='cuda'); z = torch.empty(n, device='cuda')
x.normal_(); z.normal_()
gb = n * 4 / 1e9
print('fill  %.0f GB/s' % (gb / t(lambda: y.fill_(1.0)) * 1e3))
print('copy  %.0f GB/s (r+w)' % (2 * gb / t(lambda: y.copy_(x)) * 1e3))
print('sum   %.0f GB/s (read)' % (gb / t(lambda: x.sum()) * 1e3))
print('add   %.0f GB/s (2r+1w)' % (3 * gb / t(lambda: torch.add(x, z, out=y)) * 1e3))
